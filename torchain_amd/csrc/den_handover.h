// Two workgroups of ONE launch that work on the same sequence (den_tied_mitm.hip, den_tied_planes.hip): pairing by ticket,
// and the one hand-over at which each publishes what it stored and waits for the other.
#pragma once

#include "den_tied_device.h"

namespace tc {

namespace {

struct MitmParams {
  uint32_t *sync;    // [0] ticket counter, [4 + 2 s + role] hand-over flags
  int M;             // meeting frame
  uint32_t aScr;     // 16 bytes of LDS scratch behind the kernel's layout: ticket, hand-over result
};

typedef __attribute__((address_space(1))) uint32_t gu32;

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

// ticket -> (sequence, role): whoever starts next becomes the partner of the last unpaired workgroup
__device__ __forceinline__ uint32_t take_ticket(const MitmParams &q) {
  if (threadIdx.x == 0)
    *reinterpret_cast<lds_u *>(q.aScr) = __hip_atomic_fetch_add((gu32 *)q.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*reinterpret_cast<lds_u *>(q.aScr));
}

}  // namespace

}  // namespace tc
