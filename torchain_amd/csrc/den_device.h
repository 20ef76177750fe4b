// Device helpers shared by the on-chip denominator kernels (den_kernels.hip: general graphs,
// den_tied_kernel.hip: tied graphs).
#pragma once

#include "chain_internal.h"

namespace tc {

// In-kernel phase stamps (diagnostic build only: make EXTRA=-DTC_PHASE_STAMPS).  Thread 0 of every
// wave of workgroup 0 accumulates shader cycles per phase; the totals go to a scratch area of the
// workspace that nothing else reads.  Never quote the run time of such a build.
#ifdef TC_PHASE_STAMPS
#define TC_STAMP_DECL long long st_prev = clock64(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long *wst = st_acc + 5;
#define TC_STAMP(i)                      \
  {                                      \
    const long long st_now = clock64();  \
    st_acc[i] += st_now - st_prev;       \
    st_prev = st_now;                    \
  }
#define TC_STAMP_FLUSH(ptr)                                                         \
  if (blockIdx.x == 0 && lane == 0)                                                 \
    for (int i = 0; i < 8; ++i) (ptr)[wave * 8 + i] = st_acc[i];
#define TC_WALK_ARG , long long *wst
#define TC_WALK_PASS , wst
#else
#define TC_WALK_ARG
#define TC_WALK_PASS
#define TC_STAMP_DECL
#define TC_STAMP(i)
#define TC_STAMP_FLUSH(ptr)
#endif

// Sum over the 64 lanes, returned in every lane.  DPP adds inside each row of 16 lanes, then the four row
// totals through v_readlane: ~12 instructions and no LDS round trips (__shfl_xor is ds_bpermute_b32, six
// dependent LDS latencies per reduction, and these reductions sit on the per-frame critical path).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x128>(v);  // row_ror:8   -> every lane holds its row's total
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// sum over the workgroup; `red` must not be written again before the next barrier
__device__ __forceinline__ float block_sum(float v, float *red, int wave, int lane) {
  v = wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  // the 16 wave totals: one per lane of every row of 16 lanes, summed with four DPP adds (5 instructions
  // instead of 8 LDS reads + 16 dependent adds; these reductions are on the per-frame critical path)
  static_assert(kWaves == 16, "one DPP row holds the wave totals");
  float t = red[lane & 15];
  t = dpp_add<0xB1>(t);
  t = dpp_add<0x4E>(t);
  t = dpp_add<0x124>(t);
  t = dpp_add<0x128>(t);
  return t;
}

__device__ __forceinline__ float exp_limited(float x) {
  // [K] later Kaldi: ApplyExpLimited(-30, 30); identical to the 22fbdd ApplyExp() for |y| < 30
  // compare-and-clamp (not fminf/fmaxf) so that a NaN input stays NaN and trips the objf check
  x = x < -30.0f ? -30.0f : x;
  x = x > 30.0f ? 30.0f : x;
  return __expf(x);
}

__device__ __forceinline__ float4 load_row4(const float *row, int i, int n, int vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec) {
    if (i < n) r = *reinterpret_cast<const float4 *>(row + i);
  } else {
    if (i < n) r.x = row[i];
    if (i + 1 < n) r.y = row[i + 1];
    if (i + 2 < n) r.z = row[i + 2];
    if (i + 3 < n) r.w = row[i + 3];
  }
  return r;
}

__device__ __forceinline__ void store_row4(float *row, int i, int n, int vec, float4 v) {
  if (vec) {
    if (i < n) *reinterpret_cast<float4 *>(row + i) = v;
  } else {
    if (i < n) row[i] = v.x;
    if (i + 1 < n) row[i + 1] = v.y;
    if (i + 2 < n) row[i + 2] = v.z;
    if (i + 3 < n) row[i + 3] = v.w;
  }
}

constexpr int kChunk = 8;
// gamma_t(pdf) is an occupation posterior (sum over pdfs = 1), accumulated as unsigned fixed point
// with 31 fractional bits: quantum 4.7e-10, exact (order-independent, bitwise reproducible) sums.
constexpr float kGammaScale = 2147483648.0f;
constexpr float kGammaInvScale = 1.0f / 2147483648.0f;

__device__ __forceinline__ float lds_at(const float *base, uint32_t byte_off) {
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}

__device__ __forceinline__ void gamma_add(float *GM, uint32_t byte_off, float v) {
  // integer LDS atomics run at store rate; float ones are lane-serialised on gfx950.
  // v_cvt_rpi_i32_f32 = floor(v + 0.5) in one instruction (v >= 0 here; a contribution of exactly 1.0,
  // i.e. 2^31, saturates to 2^31 - 1: one quantum)
  int32_t q;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(v));
  atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(GM) + byte_off), (uint32_t)q);
}

}  // namespace tc
