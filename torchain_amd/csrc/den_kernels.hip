// Fused denominator forward-backward for gfx950 (MI355X).
//
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), the
// part of the reference's hot call (src/my_lib_chain.cpp:129-131) that the headline metric times.
//
// Mapping.  The frame recursion is serial and sequences never interact, so one workgroup (16 waves,
// one per CU) owns one sequence for all 2T frames and keeps that sequence's per-frame working set in
// LDS: exp(y_t), alpha'_t / beta_{t+1} (gather source), one accumulator per row and, in the backward
// half, gamma_t (u32 fixed point) and alpha'_t.  HBM sees each y row twice (forward, backward), each
// alpha' frame once out and once back, and each derivative row once.  The transition tables are
// streamed from L2 as a per-wave cell stream in a lane-major schedule (schedule_*.cpp): a lane walks one
// state's arc list, a wave instruction loads 64 lanes x 16 contiguous bytes, and the lanes gather
// alpha' (and exp(y)) from LDS.  There are no LDS float atomics (192 cycles per wave-instruction on
// gfx950): every row sum is committed with a plain store and gamma is integer fixed point.
// Two code paths: the general kernel (any graph; two gathers per arc, one gamma atomic per arc in the
// backward walk, in-band ROW cells, three barriers per frame) and the "tied" kernel for chain-structured
// graphs (all non-self-loop arcs into a state share a pdf): one gather per arc in both walks, no atomics
// in the walks, self-loops and gamma handled per state, and OWNER-COMPUTES schedules -- the thread that
// owns a state walks its arc list, so row sums never cross threads and a frame needs two barriers.
//
// Numerics follow the Kaldi CPU arithmetic: linear domain, fp32, per-frame renormalisation by the
// alpha-sum of the previous frame ("arbitrary_scale"), leaky-HMM mixing, betas carrying 1/tot_prob.
#include "den_device.h"

namespace tc {

// ---- general graphs: walk over 8-byte cells with in-band ROW cells ------------------------------------
// One chunk of kChunk cells of a wave's stream.  All gathers of the chunk are issued before the first
// use (ROW cells carry valid offsets, so their gathers are harmless); only the commit of a finished
// row sits behind a branch, and that branch is scalar (the ROW flag is the same in all 64 lanes).
//   FWD: acc(row) += alpha'(src) * w * p(pdf)
//   BWD: vf = w * beta(dst) * p(pdf); acc(row) += vf; gamma(pdf) += vf * alpha'(row) / alpha_sum
// Folds the private slots of split rows into their state's accumulator; called by the thread that
// owns the state, after the barrier that follows the arc walk and before it reads the sum.
__device__ __forceinline__ void fold_split_rows(const ScheduleDev &sc, int fx0, int fx1, float *ACC) {
  for (int e = fx0; e < fx1; ++e) {
    const int2 f = sc.fix[e];
    ACC[f.x] += ACC[f.y];
  }
}

struct RowState {
  uint32_t row;
  float acc, occf;
};

template <bool BWD, bool ALPHA_LDS>
__device__ __forceinline__ void process_chunk(const uint2 (&q)[kChunk], RowState &rs,
                                              const float *__restrict__ SRC, const float *__restrict__ PB,
                                              float *__restrict__ ACC, float *__restrict__ GM,
                                              const float *__restrict__ AL, const float *__restrict__ hist_t,
                                              float inv_asum) {
  float a[kChunk], pp[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    a[u] = lds_at(SRC, q[u].y >> 16);
    pp[u] = lds_at(PB, q[u].y & 0xfffcu);
  }
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    float w = __uint_as_float(q[u].x);
    // the ROW flag is in-band (bit 0 of the offset word, identical in all 64 lanes): testing the cell
    // itself avoids a separate mask stream and its dependent L2 load in every loop iteration
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(q[u].y) & 1u, 0)) {  // rare, scalar branch
      ACC[rs.row] = rs.acc;  // every row owns its slot: plain store, no atomic
      rs.acc = 0.f;
      rs.row = q[u].x & 0xffffu;  // ROW cell x = slot | state << 16
      if (BWD) {
        if (ALPHA_LDS)
          rs.occf = lds_at(AL, (q[u].x >> 14) & 0x3fffcu) * inv_asum * kGammaScale;
        else
          rs.occf = hist_t[q[u].x >> 16] * inv_asum * kGammaScale;
      }
      w = 0.f;  // the arc math below then adds nothing for this cell
    }
    if (!BWD) {
      rs.acc = fmaf(a[u] * w, pp[u], rs.acc);
    } else {
      const float vf = w * a[u] * pp[u];
      rs.acc += vf;
      gamma_add(GM, q[u].y & 0xfffcu, vf * rs.occf);  // a ROW cell adds 0 at its lane-aligned dummy offset
    }
  }
}

// Consumes this wave's cell stream with two register buffers in ping-pong: while one chunk is being
// processed the next one (4 KB per wave, 64 KB per CU) is in flight from L2.  The row mask of the NEXT
// iteration is fetched at the top of this one and only made scalar at the bottom, so its L2 round trip
// is never waited for on its own (vmcnt retires in order: waiting for it early would drain the
// prefetched chunk as well).
template <bool BWD, bool ALPHA_LDS>
__device__ __forceinline__ void walk_rows(const ScheduleDev &sc, int wave, int lane, int dummy_row,
                                          const float *__restrict__ SRC, const float *__restrict__ PB,
                                          float *__restrict__ ACC, float *__restrict__ GM,
                                          const float *__restrict__ AL, const float *__restrict__ hist_t,
                                          float inv_asum) {
  const int2 range = sc.wave_range[wave];
  const int first = __builtin_amdgcn_readfirstlane(range.x);   // multiple of kStreamUnroll
  const int ncells = __builtin_amdgcn_readfirstlane(range.y);  // multiple of kStreamUnroll
  // cells are stored [pair][lane][2]: one 16-byte load brings a lane's cells 2p and 2p+1
  const uint4 *__restrict__ r = reinterpret_cast<const uint4 *>(sc.cells) + (int64_t)(first / 2) * 64 + lane;
  uint2 qa[kChunk], qb[kChunk];
  auto load_chunk = [&](uint2 (&q)[kChunk], int cell0) {
#pragma unroll
    for (int u = 0; u < kChunk / 2; ++u) {
      const uint4 v = r[(cell0 / 2 + u) * 64];
      q[2 * u] = make_uint2(v.x, v.y);
      q[2 * u + 1] = make_uint2(v.z, v.w);
    }
  };
  load_chunk(qa, 0);
  RowState rs;
  rs.row = (uint32_t)dummy_row;  // until the stream's first ROW cell
  rs.acc = 0.f;
  rs.occf = 0.f;
  for (int c = 0; c < ncells; c += kStreamUnroll) {
    // Self-balancing: a wave's issue priority falls as it advances through its stream, so the waves
    // that lag (the arbiter otherwise favours the oldest) outrank the leaders and all 16 reach the
    // barrier together instead of leaving a tail with few active waves.
    if (4 * c < ncells)
      __builtin_amdgcn_s_setprio(3);
    else if (2 * c < ncells)
      __builtin_amdgcn_s_setprio(2);
    else if (4 * c < 3 * ncells)
      __builtin_amdgcn_s_setprio(1);
    else
      __builtin_amdgcn_s_setprio(0);
    // the stream is followed by kStreamUnroll readable padding cells, so these loads need no guard
    load_chunk(qb, c + kChunk);
    process_chunk<BWD, ALPHA_LDS>(qa, rs, SRC, PB, ACC, GM, AL, hist_t, inv_asum);
    load_chunk(qa, c + kStreamUnroll);
    process_chunk<BWD, ALPHA_LDS>(qb, rs, SRC, PB, ACC, GM, AL, hist_t, inv_asum);
  }
  __builtin_amdgcn_s_setprio(0);
}

// ---- tied graphs: owner-computes walk over 6-byte cells ---------------------------------------------
// A pair of cells is {w0, w1, off0 | off1 << 16}: fp32 weights and 16-bit LDS byte offsets of the two
// gathers.  Both tied walks are the same operation -- acc(row) += w * SRC[off] -- with SRC = alpha'_t
// (forward) or Y_t (backward), and the row a lane is summing is always one of its OWN states (or a
// secondary row, see schedule_owner.cpp build_owner): the sum is committed to the thread's own accumulator
// slot and read back by the same thread, so no barrier separates the walk from the per-state pass.
struct Pair6 {
  uint32_t w0, w1, off;
};

// LDS access by absolute byte address: the kernel has no static __shared__, so the dynamic LDS block
// starts at address 0 and the compiler need not add a (relocated) base to every gather address.
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ float lds_abs(uint32_t byte_addr) { return *reinterpret_cast<lds_float *>(byte_addr); }
__device__ __forceinline__ void lds_abs_store(uint32_t byte_addr, float v) {
  *reinterpret_cast<lds_float *>(byte_addr) = v;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// Where the row a wave is currently summing commits: all 64 lanes are at the same row index k.
struct RowCursor {
  uint32_t base;   // per lane: own accumulator slot 0 (k < K), then the wave's first secondary slot
  uint32_t koff;   // uniform: byte offset of row k from `base`
  int k;           // uniform
  uint32_t fix_base;
  int K;
  __device__ __forceinline__ void advance() {
    ++k;
    if (k < K) {  // own rows: k -> 4 * (tid + 1024 * (k >> 2)) + (k & 3)
      koff += 4u;
      if ((koff & 12u) == 0u) koff += 4u * 4u * kThreads - 16u;
    } else if (k == K) {
      base = fix_base;
      koff = 0u;
    } else {
      koff += 256u;  // secondary rows: 64 consecutive private slots per row
    }
  }
};

// One chunk = 4 pairs.  Row ends are wave-uniform and come from the schedule's mask words through the
// scalar cache: bit u of `mb` <=> a row ends with the SECOND cell of pair u, bit u of `ma` <=> with its
// FIRST cell (the pair straddles two rows: rows are not padded to whole pairs).  Commits sit behind
// scalar branches; the common path is two gathers and one packed FMA (v_pk_fma_f32) per pair.
template <uint32_t SRC_BASE>
__device__ __forceinline__ void process_chunk6(const Pair6 (&q)[kChunk / 2], uint32_t mb, uint32_t ma, RowCursor &rc,
                                               v2f &acc) {
  v2f a[kChunk / 2];
#pragma unroll
  for (int u = 0; u < kChunk / 2; ++u) {
#ifdef TC_ABL_NOGATHER
    a[u].x = __uint_as_float(q[u].off & 0xffffu);
    a[u].y = __uint_as_float(q[u].off >> 16);
#else
    a[u].x = lds_abs(SRC_BASE + (q[u].off & 0xffffu));
    a[u].y = lds_abs(SRC_BASE + (q[u].off >> 16));
#endif
  }
#pragma unroll
  for (int u = 0; u < kChunk / 2; ++u) {
    v2f w;
    w.x = __uint_as_float(q[u].w0);
    w.y = __uint_as_float(q[u].w1);
    if (__builtin_expect((ma >> u) & 1u, 0)) {
      // the first cell closes the current row, the second opens the next one
      lds_abs_store(rc.base + rc.koff, fmaf(a[u].x, w.x, acc.x) + acc.y);
      rc.advance();
      acc.x = 0.f;
      acc.y = a[u].y * w.y;
    } else {
      acc = __builtin_elementwise_fma(a[u], w, acc);
    }
    if (__builtin_expect((mb >> u) & 1u, 0)) {
      lds_abs_store(rc.base + rc.koff, acc.x + acc.y);  // commit the finished row
      acc.x = 0.f;
      acc.y = 0.f;
      rc.advance();
    }
  }
}

// A wave's share of a frame is only a handful of chunks, so the L2 round trip of the first one is a
// visible fraction of the walk: the kernel issues it BEFORE the frame's barrier (the stream is the same
// every frame) and the wait overlaps the latency.
__device__ __forceinline__ const uint4 *walk6_base(const ScheduleDev &sc, int wave, int lane, int &ncells) {
  const int2 range = sc.wave_range[wave];
  const int first = __builtin_amdgcn_readfirstlane(range.x);  // multiple of kChunk
  ncells = __builtin_amdgcn_readfirstlane(range.y);           // multiple of kChunk
  // the stream is stored [chunk of 8 cells][3 blocks][lane]{16 bytes}: see schedule_owner.cpp
  return reinterpret_cast<const uint4 *>(sc.cells) + (int64_t)(first / kChunk) * 3 * 64 + lane;
}

__device__ __forceinline__ void load_chunk6(Pair6 (&q)[kChunk / 2], const uint4 *__restrict__ r, int cell0) {
#ifdef TC_ABL_SAME
  cell0 = 0;
#endif
  const uint4 *rc = r + (int64_t)(cell0 / kChunk) * 3 * 64;
  const uint4 wa = rc[0], wb = rc[64], oc = rc[128];
  q[0] = Pair6{wa.x, wa.y, oc.x};
  q[1] = Pair6{wa.z, wa.w, oc.y};
  q[2] = Pair6{wb.x, wb.y, oc.z};
  q[3] = Pair6{wb.z, wb.w, oc.w};
}

// Two register buffers in ping-pong; qa arrives preloaded with the first STREAMED chunk.  The stream and
// the mask words are followed by readable padding, so the loads past the wave's range need no guard.
// RES = 2: the wave's first two chunks (one mask word) are held in registers by the caller for the whole
// phase (ra, rb) and never re-read: the walk is bound by the L2 -> CU stream path, so every resident
// chunk is time saved.  Every wave's range is at least two chunks long (schedule_owner.cpp).
#ifdef TC_PHASE_STAMPS
// diagnostic: cycles spent waiting for the current chunk's loads (three younger loads may stay in
// flight) and cycles spent processing it, accumulated into stamp slots 5 and 6
#define TC_WALK_WAIT                                  \
  {                                                   \
    const long long w0 = clock64();                   \
    __builtin_amdgcn_s_waitcnt(0x0F73); /* vmcnt(3) */ \
    wst[0] += clock64() - w0;                         \
    wst[2] = clock64();                               \
  }
#define TC_WALK_PROC                                  \
  {                                                   \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ \
    wst[1] += clock64() - wst[2];                     \
  }
#define TC_WALK_ARG , long long *wst
#define TC_WALK_PASS , wst
#else
#define TC_WALK_WAIT
#define TC_WALK_PROC
#define TC_WALK_ARG
#define TC_WALK_PASS
#endif

template <uint32_t SRC_BASE, int RES>
__device__ __forceinline__ void walk_rows6(const uint4 *__restrict__ r, int ncells,
                                           const uint32_t *__restrict__ masks, Pair6 (&qa)[kChunk / 2],
                                           RowCursor rc, const Pair6 (&ra)[kChunk / 2],
                                           const Pair6 (&rb)[kChunk / 2] TC_WALK_ARG) {
  static_assert(RES == 0 || RES == 2, "resident prefix is zero or two chunks");
  Pair6 qb[kChunk / 2];
  v2f acc = {0.f, 0.f};
  // constant address space + uniform address = s_load_dword: the mask words never touch the vector
  // memory pipe (a vector load here would also serialise behind every older load: vmcnt is in-order)
  typedef __attribute__((address_space(4))) const uint32_t const_u32;
  const_u32 *mk = (const_u32 *)masks;
  uint32_t m = mk[0];
  if (RES == 2) {
    const uint32_t m1 = mk[1];
    process_chunk6<SRC_BASE>(ra, m, m >> 8, rc, acc);
    process_chunk6<SRC_BASE>(rb, m >> 4, m >> 12, rc, acc);
    m = m1;
  }
  for (int c = RES * kChunk; c < ncells; c += 2 * kChunk) {
    // (no s_setprio here: the progress-based priorities of the general walk cost 1 % on this one)
    load_chunk6(qb, r, c + kChunk);
    const uint32_t mnext = mk[(c >> 4) + 1];
    TC_WALK_WAIT
    process_chunk6<SRC_BASE>(qa, m, m >> 8, rc, acc);
    TC_WALK_PROC
    if (c + kChunk >= ncells) break;
    load_chunk6(qa, r, c + 2 * kChunk);
    TC_WALK_WAIT
    process_chunk6<SRC_BASE>(qb, m >> 4, m >> 12, rc, acc);
    TC_WALK_PROC
    m = mnext;
  }
}

// Tied graphs: per-state self-loop and forward-pdf terms, applied by the thread that owns the state.
//   fs = forward-pdf*4 | self-loop-pdf*4 << 16 (LDS byte offsets into exp(y)), ws = self-loop prob.
__device__ __forceinline__ float tied_alpha(const float *PB, uint32_t fs, float ws, float F, float a_self) {
  // alpha_{t+1}(g) * asum_t = p(f(g)) * sum_{h != g} w * alpha'_t(h)  +  p(s(g)) * w_s * alpha'_t(g)
  return fmaf(lds_at(PB, fs & 0xffffu), F, lds_at(PB, fs >> 16) * (ws * a_self));
}

// The CU serves older waves first wherever waves contend, so the youngest wave of each SIMD finishes its
// walk last and every frame waits for it.  During the tied walks the four wave generations therefore run
// at issue priorities 0..3, youngest highest (-1.5 % run time; keeping the priority through the per-state
// pass as well is worse).
#define TC_AGE_PRIO_ON                                            \
  {                                                               \
    if (wave >= 12) __builtin_amdgcn_s_setprio(3);                \
    else if (wave >= 8) __builtin_amdgcn_s_setprio(2);            \
    else if (wave >= 4) __builtin_amdgcn_s_setprio(1);            \
  }
#define TC_AGE_PRIO_OFF __builtin_amdgcn_s_setprio(0);

template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, bool WANT_DERIV, bool TIED>
__global__ __launch_bounds__(kThreads) void den_fwd_bwd_kernel(const DenParams p) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int H = p.H, P = p.P, S = p.S, T = p.T;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  // Which of its JV float4s of states a thread really has: tied graphs are laid out in whole planes of
  // 4096 positions (schedule_owner.cpp build_owner), so the test is wave-uniform there (a scalar branch, no
  // per-lane compare and exec masking in the per-state passes).
  const int planes = Hs / (4 * kThreads);
  auto owns = [&](int j, int h0) { return TIED ? j < planes : h0 < Hs; };
  float *const PB = lds;                          // exp(y_t), at LDS offset 0
  float *const A0 = lds + PV * 4 * kThreads;      // alpha'_t (forward) / beta_{t+1} (backward): gather source
  float *const ACC = lds + p.L.off_acc; // row accumulators: one per state, dummy at Hs, then split-row slots
  float *const GM = lds + p.L.off_g;    // gamma_t as u32 fixed point (backward only)
  float *const AL = lds + p.L.off_al;   // alpha'_t (backward only, when it fits)
  float *const red = lds + p.L.off_red;
  float *const asum_h = lds + p.L.off_asum;  // alpha-sum of every frame

  float4 cpi[JV];  // leaky * pi for the states this thread owns
  float4 pi4[JV];
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    const int h0 = 4 * (tid + kThreads * j);
    pi4[j] = owns(j, h0) ? *reinterpret_cast<const float4 *>(p.pi + h0) : make_float4(0.f, 0.f, 0.f, 0.f);
    cpi[j] = make_float4(p.leaky * pi4[j].x, p.leaky * pi4[j].y, p.leaky * pi4[j].z, p.leaky * pi4[j].w);
    part += (pi4[j].x + pi4[j].y) + (pi4[j].z + pi4[j].w);
  }
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  float asum = block_sum(part, red, wave, lane);
  float *hist = p.alpha_hist + (int64_t)s * Hs;  // frame t lives at hist + t*S*Hs
  const int64_t hist_step = (int64_t)S * Hs;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    const int h0 = 4 * (tid + kThreads * j);
    if (owns(j, h0)) {
      float4 a = make_float4(pi4[j].x + cpi[j].x * asum, pi4[j].y + cpi[j].y * asum, pi4[j].z + cpi[j].z * asum,
                             pi4[j].w + cpi[j].w * asum);
      *reinterpret_cast<float4 *>(A0 + h0) = a;
      *reinterpret_cast<float4 *>(hist + h0) = a;
      *reinterpret_cast<float4 *>(ACC + h0) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (tid < 4) ACC[Hs + tid] = 0.f;
  float y2 = 0.f;
  {
    const float *yrow = p.y + (int64_t)s * p.y_stride;
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      if (i0 < Ps) {
        float4 yv = load_row4(yrow, i0, P, p.y_vec);
        y2 += (yv.x * yv.x + yv.y * yv.y) + (yv.z * yv.z + yv.w * yv.w);
        *reinterpret_cast<float4 *>(PB + i0) =
            make_float4(exp_limited(yv.x), exp_limited(yv.y), exp_limited(yv.z), exp_limited(yv.w));
      }
    }
  }
  double logsum = 0.0;  // thread 0 only
  const int ffx0 = p.fwd.fix_begin[tid], ffx1 = p.fwd.fix_begin[tid + 1];
  if (tid == 0) asum_h[0] = asum;
  float inv_prev = 1.0f / asum;
  float asum_prev = asum;
  int fwd_n = 0, bwd_n = 0;
  const uint4 *fwd_r = nullptr, *bwd_r = nullptr;
  const uint32_t *fwd_m = nullptr, *bwd_m = nullptr;
  RowCursor fwd_rc, bwd_rc;
  // forward stream, chunks 0 and 1: register-resident for the forward phase when a thread owns two float4s of
  // states; with four, the registers are better spent on the per-state arrays
  constexpr int kFwdRes = JV <= 2 ? 2 : 0;
  Pair6 fres0[kChunk / 2], fres1[kChunk / 2];
  if (TIED) {
    fwd_r = walk6_base(p.fwd, wave, lane, fwd_n);
    if (kFwdRes) {
      load_chunk6(fres0, fwd_r, 0);
      load_chunk6(fres1, fwd_r, kChunk);
    }
    bwd_r = walk6_base(p.bwd, wave, lane, bwd_n);
    fwd_m = p.fwd.masks + wave * p.fwd.mask_stride;
    bwd_m = p.bwd.masks + wave * p.bwd.mask_stride;
    const uint32_t acc0 = (uint32_t)p.L.off_acc * 4u;
    const uint32_t own = acc0 + 16u * (uint32_t)tid;
    const int K = Hs / kThreads;
    fwd_rc = RowCursor{own, 0u, 0, acc0 + 4u * (uint32_t)(Hs + 4 + 64 * p.fwd.extra_first[wave] + lane), K};
    bwd_rc = RowCursor{own, 0u, 0, acc0 + 4u * (uint32_t)(Hs + 4 + 64 * p.bwd.extra_first[wave] + lane), K};
  }

  // ---- forward frames t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  TC_STAMP_DECL
  for (int t = 1; t <= T; ++t) {
    TC_STAMP(0)
    Pair6 q0[kChunk / 2];
    if (TIED) load_chunk6(q0, fwd_r, kFwdRes * kChunk);
    __syncthreads();  // A0, PB, ACC ready
    TC_STAMP(1)
    float4 yreg[PV];
    {
      // prefetch y_t under the arc walk (the last iteration re-reads row T-1: keeps yreg in registers)
      const float *yrow = p.y + ((int64_t)(t < T ? t : T - 1) * S + s) * p.y_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) yreg[v] = load_row4(yrow, 4 * (tid + kThreads * v), P, p.y_vec);
    }
    if (TIED)
    {
      TC_AGE_PRIO_ON
      walk_rows6<PV * 16 * kThreads, kFwdRes>(fwd_r, fwd_n, fwd_m, q0, fwd_rc, fres0, fres1 TC_WALK_PASS);
      TC_AGE_PRIO_OFF
    }
    else
      walk_rows<false, true>(p.fwd, wave, lane, Hs, A0, PB, ACC, nullptr, nullptr, nullptr, 0.f);
    TC_STAMP(2)
    if (!TIED || p.fwd.nfix) {
      __syncthreads();  // all row sums committed (tied graphs: only when rows were split)
      fold_split_rows(p.fwd, ffx0, ffx1, ACC);
    }
    TC_STAMP(3)
    float4 v4[JV];
    part = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      v4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (owns(j, h0)) {
        float4 a = *reinterpret_cast<float4 *>(ACC + h0);
        if (TIED) {
          const uint4 fs = *reinterpret_cast<const uint4 *>(p.tied_fs + h0);
          const float4 ws = *reinterpret_cast<const float4 *>(p.tied_w + h0);
          const float4 al = *reinterpret_cast<float4 *>(A0 + h0);  // alpha'_t of the owned states
          a = make_float4(tied_alpha(PB, fs.x, ws.x, a.x, al.x), tied_alpha(PB, fs.y, ws.y, a.y, al.y),
                          tied_alpha(PB, fs.z, ws.z, a.z, al.z), tied_alpha(PB, fs.w, ws.w, a.w, al.w));
        }
        v4[j] = make_float4(a.x * inv_prev, a.y * inv_prev, a.z * inv_prev, a.w * inv_prev);
        part += (v4[j].x + v4[j].y) + (v4[j].z + v4[j].w);
      }
    }
    asum = block_sum(part, red, wave, lane);
    TC_STAMP(4)
    float *hist_t = hist + (int64_t)t * hist_step;
    float part_tot = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      if (owns(j, h0)) {
        float4 a = make_float4(v4[j].x + cpi[j].x * asum, v4[j].y + cpi[j].y * asum, v4[j].z + cpi[j].z * asum,
                               v4[j].w + cpi[j].w * asum);
        *reinterpret_cast<float4 *>(A0 + h0) = a;
        *reinterpret_cast<float4 *>(hist_t + h0) = a;
        part_tot += (a.x + a.y) + (a.z + a.w);
      }
    }
    if (t < T) {
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps) {
          float4 yv = yreg[v];
          y2 += (yv.x * yv.x + yv.y * yv.y) + (yv.z * yv.z + yv.w * yv.w);
          *reinterpret_cast<float4 *>(PB + i0) =
              make_float4(exp_limited(yv.x), exp_limited(yv.y), exp_limited(yv.z), exp_limited(yv.w));
        }
      }
    }
    if (tid == 0) {
      asum_h[t] = asum;
      logsum += (double)__logf(asum_prev);  // log of the scale divided out of frame t
    }
    asum_prev = asum;
    inv_prev = 1.0f / asum;
    if (t == T) part = part_tot;
  }
  TC_STAMP(0)
  TC_STAMP_FLUSH(p.stamps)
  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  const float tot = block_sum(part, red + kWaves, wave, lane);
  {
    const double y2d = (double)block_sum(y2, red + 2 * kWaves, wave, lane);
    if (tid == 0) {
      p.seq_logprob[s] = logsum + (double)__logf(tot);
      p.seq_y2[s] = y2d;
    }
  }
  if (!WANT_DERIV) return;

  // ---- backward   ([K] BetaDashLastFrame, Beta(T), then BetaDashGeneralFrame(t) + Beta(t))
  // beta'_T(h) = 1/tot;  beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The same LDS regions now
  // hold B (= beta_{t+1}), BACC, P, GAMMA and alpha'_t.
  const float inv_tot = 1.0f / tot;
  part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) part += ((cpi[j].x + cpi[j].y) + (cpi[j].z + cpi[j].w)) * inv_tot;
  float bsum = block_sum(part, red + 3 * kWaves, wave, lane);  // also orders the A0 reuse below
  float4 areg[JV];
  float4 ycur[PV], ynext[PV];
  float4 bown[JV];  // tied graphs: beta_{t+1} of the owned states (the LDS gather source holds Y instead)
  const int bfx0 = p.bwd.fix_begin[tid], bfx1 = p.bwd.fix_begin[tid + 1];
  // tied graphs keep two exp(y) buffers in the backward pass: frame t (self-loop terms of the owner
  // pass) and frame t-1 (written under the arc walk, needed to form Y for the next frame)
  float *PBcur = PB, *PBnext = (TIED && !ALPHA_LDS) ? PB : lds + p.L.off_p2;  // tight tied layout: one exp(y) buffer
  {
    const float *hist_t = hist + (int64_t)(T - 1) * hist_step;
    const float *yrow = p.y + ((int64_t)(T - 1) * S + s) * p.y_stride;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      bown[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (owns(j, h0)) {
        float4 b = make_float4(h0 < H ? inv_tot + bsum : 0.f, h0 + 1 < H ? inv_tot + bsum : 0.f,
                               h0 + 2 < H ? inv_tot + bsum : 0.f, h0 + 3 < H ? inv_tot + bsum : 0.f);
        bown[j] = b;
        if (!TIED) *reinterpret_cast<float4 *>(A0 + h0) = b;
        *reinterpret_cast<float4 *>(ACC + h0) = make_float4(0.f, 0.f, 0.f, 0.f);  // states with no out-arcs
        if (TIED) {  // alpha'_{t+1} of the owned states is parked in the thread's own AL slots (roomy layout)
          if (ALPHA_LDS)
            *reinterpret_cast<float4 *>(AL + h0) = *reinterpret_cast<const float4 *>(hist_t + hist_step + h0);
        } else if (ALPHA_LDS) {
          *reinterpret_cast<float4 *>(AL + h0) = *reinterpret_cast<const float4 *>(hist_t + h0);
        }
      }
    }
    if (!TIED && ALPHA_LDS && tid < 4) AL[Hs + tid] = 0.f;
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      ycur[v] = load_row4(yrow, i0, P, p.y_vec);
      if (i0 < Ps) {
        *reinterpret_cast<float4 *>(PBcur + i0) = make_float4(exp_limited(ycur[v].x), exp_limited(ycur[v].y),
                                                              exp_limited(ycur[v].z), exp_limited(ycur[v].w));
        *reinterpret_cast<float4 *>(GM + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (TIED) {
      __syncthreads();  // exp(y_{T-1}) complete: Y_{T-1}(g) = beta_T(g) * p_{T-1}(f(g))
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        const int h0 = 4 * (tid + kThreads * j);
        if (owns(j, h0)) {
          const uint4 fs = *reinterpret_cast<const uint4 *>(p.tied_fs + h0);
          *reinterpret_cast<float4 *>(A0 + h0) =
              make_float4(bown[j].x * lds_at(PBcur, fs.x & 0xffffu), bown[j].y * lds_at(PBcur, fs.y & 0xffffu),
                          bown[j].z * lds_at(PBcur, fs.z & 0xffffu), bown[j].w * lds_at(PBcur, fs.w & 0xffffu));
        }
      }
    }
  }
  TC_STAMP(0)
#ifdef TC_PHASE_STAMPS
  for (int i = 0; i < 8; ++i) st_acc[i] = 0;
#endif
  for (int t = T - 1; t >= 0; --t) {
    TC_STAMP(0)
    Pair6 q0[kChunk / 2];
    if (TIED) load_chunk6(q0, bwd_r, 0);
    __syncthreads();  // B (or Y), PB, AL ready; BACC and GAMMA zero
    TC_STAMP(1)
    const float asum_t = asum_h[t];
    const float inv_as = 1.0f / asum_t;
    const float *hist_t = hist + (int64_t)t * hist_step;
    {
      // prefetch frame t-1 (y row and alpha') under the arc walk; at t == 0 it re-reads frame 0
      const int tn = t > 0 ? t - 1 : 0;
      const float *yrow = p.y + ((int64_t)tn * S + s) * p.y_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) ynext[v] = load_row4(yrow, 4 * (tid + kThreads * v), P, p.y_vec);
      if (ALPHA_LDS || TIED) {
        // general graphs: alpha'_{t-1} for the next frame's LDS copy; tied graphs: alpha'_t of the owned
        // states for THIS frame's per-state pass (the walk hides the latency; no second register set)
        const float *hist_n = hist + (int64_t)(TIED ? t : tn) * hist_step;
#pragma unroll
        for (int j = 0; j < JV; ++j) {
          const int h0 = 4 * (tid + kThreads * j);
          areg[j] = owns(j, h0) ? *reinterpret_cast<const float4 *>(hist_n + h0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    if (TIED)  // beta'_t(h) * asum_t = sum over out-arcs of w * Y(dst): the same walk as forward, no atomics
    {
      TC_AGE_PRIO_ON
      walk_rows6<PV * 16 * kThreads, 0>(bwd_r, bwd_n, bwd_m, q0, bwd_rc, q0, q0 TC_WALK_PASS);
      TC_AGE_PRIO_OFF
    }
    else
      walk_rows<true, ALPHA_LDS>(p.bwd, wave, lane, Hs, A0, PBcur, ACC, GM, AL, hist_t, inv_as);
    if (TIED && ALPHA_LDS) {
      // exp(y_{t-1}) into the other buffer while the slower waves finish their walk
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps)
          *reinterpret_cast<float4 *>(PBnext + i0) = make_float4(exp_limited(ynext[v].x), exp_limited(ynext[v].y),
                                                                 exp_limited(ynext[v].z), exp_limited(ynext[v].w));
      }
    }
    TC_STAMP(2)
    if (!TIED || p.bwd.nfix) {
      __syncthreads();  // beta' sums and gamma committed (tied graphs: only when rows were split)
      fold_split_rows(p.bwd, bfx0, bfx1, ACC);
    }
    TC_STAMP(3)
    float4 b4[JV];
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
    // all of the thread's table loads first: one L2 round trip per frame, not one per float4 of states
    uint4 bfs[JV];
    float4 bws[JV];
    // (only worth its registers with two float4s of states and at most two of pdfs per thread: beyond, the
    // hoisted tables spill)
    constexpr bool kHoistTables = JV <= 2 && PV <= 2;
    if (TIED && kHoistTables) {
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        const int h0 = 4 * (tid + kThreads * j);
        if (owns(j, h0)) {
          bfs[j] = *reinterpret_cast<const uint4 *>(p.tied_fs + h0);
          bws[j] = *reinterpret_cast<const float4 *>(p.tied_w + h0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (owns(j, h0)) {
        float4 a = *reinterpret_cast<float4 *>(ACC + h0);
        float4 al = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TIED)
          al = areg[j];
        else if (t == 0)
          al = ALPHA_LDS ? *reinterpret_cast<float4 *>(AL + h0) : *reinterpret_cast<const float4 *>(hist_t + h0);
        if (TIED) {
          // Everything the arcs INTO an owned state g contribute to gamma_t, from per-state quantities:
          //   self-loop arc : occ_s = w_s * beta_{t+1}(g) * p_t(s(g)) * alpha'_t(g) / asum_t   -> gamma_t(s(g))
          //   forward class : sum_h w alpha'_t(h) p_t(f(g)) / asum_t = alpha_{t+1}(g) - selfpart, so
          //                   occ_f = beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart)               -> gamma_t(f(g))
          // with alpha_{t+1} = alpha'_{t+1} - leaky*pi*asum_{t+1} from the history.  The self-loop arc also
          // adds vf_s = w_s * beta_{t+1}(g) * p_t(s(g)) to beta'_t(g) * asum_t.
          const uint4 fs = kHoistTables ? bfs[j] : *reinterpret_cast<const uint4 *>(p.tied_fs + h0);
          const float4 ws = kHoistTables ? bws[j] : *reinterpret_cast<const float4 *>(p.tied_w + h0);
          const float asum_up = asum_h[t + 1];
          // alpha'_{t+1}: parked by this thread (roomy layout) or re-read from the history (tight layout)
          const float4 aup = ALPHA_LDS ? *reinterpret_cast<float4 *>(AL + h0)
                                       : *reinterpret_cast<const float4 *>(hist_t + hist_step + h0);
          auto one = [&](uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx, float &ax) {
            const float ps_ws = lds_at(PBcur, fsx >> 16) * wsx;
            const float selfpart = ps_ws * alx * inv_as;           // self-loop part of alpha_{t+1}(g)
            const float alpha_up = aupx - cpx * asum_up;            // alpha_{t+1}(g)
            ax += ps_ws * bo;                                       // vf_s into beta'_t(g) * asum_t
            const float bos = kGammaScale * bo;  // power-of-two scale: exact
            gamma_add(GM, fsx >> 16, bos * selfpart);
            gamma_add(GM, fsx & 0xffffu, bos * fmaxf(alpha_up - selfpart, 0.f));
          };
          one(fs.x, ws.x, bown[j].x, al.x, aup.x, cpi[j].x, a.x);
          one(fs.y, ws.y, bown[j].y, al.y, aup.y, cpi[j].y, a.y);
          one(fs.z, ws.z, bown[j].z, al.z, aup.z, cpi[j].z, a.z);
          one(fs.w, ws.w, bown[j].w, al.w, aup.w, cpi[j].w, a.w);
        }
        b4[j] = make_float4(a.x * inv_as, a.y * inv_as, a.z * inv_as, a.w * inv_as);  // [K] * inv_arbitrary_scale
        part += (cpi[j].x * b4[j].x + cpi[j].y * b4[j].y) + (cpi[j].z * b4[j].z + cpi[j].w * b4[j].w);
        if (t == 0) part_ab += (al.x * b4[j].x + al.y * b4[j].y) + (al.z * b4[j].z + al.w * b4[j].w);
      }
    }
    bsum = block_sum(part, red, wave, lane);  // its barrier also completes gamma_t (owner-side adds of tied graphs)
    TC_STAMP(4)
    {
      float *drow = p.deriv + ((int64_t)t * S + s) * p.deriv_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps) {
          const uint4 gu = *reinterpret_cast<uint4 *>(GM + i0);
          *reinterpret_cast<float4 *>(GM + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 g = make_float4((float)gu.x * kGammaInvScale, (float)gu.y * kGammaInvScale,
                                       (float)gu.z * kGammaInvScale, (float)gu.w * kGammaInvScale);
          if (t == 0) part_g += (g.x + g.y) + (g.z + g.w);
          float4 o = make_float4(p.deriv_weight * g.x - p.l2_scale * ycur[v].x,
                                 p.deriv_weight * g.y - p.l2_scale * ycur[v].y,
                                 p.deriv_weight * g.z - p.l2_scale * ycur[v].z,
                                 p.deriv_weight * g.w - p.l2_scale * ycur[v].w);
          if (ACCUM) {
            float4 old = load_row4(drow, i0, P, p.d_vec);
            o = make_float4(old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w);
          }
          store_row4(drow, i0, P, p.d_vec, o);
        }
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum(part_ab, red + kWaves, wave, lane);
      const float gs = block_sum(part_g, red + 2 * kWaves, wave, lane);
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
    if (TIED && !ALPHA_LDS) {
      // tight layout: exp(y_{t-1}) overwrites exp(y_t) in place -- its readers (the per-state pass) are
      // behind the reduction's barrier -- and one more barrier publishes it to the Y update below
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps)
          *reinterpret_cast<float4 *>(PB + i0) = make_float4(exp_limited(ynext[v].x), exp_limited(ynext[v].y),
                                                             exp_limited(ynext[v].z), exp_limited(ynext[v].w));
      }
      __syncthreads();
    }
    // beta_t = beta'_t + leaky-sum; next frame's gather source (beta_t, or Y_{t-1} = beta_t * p_{t-1}(f) when tied)
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      if (owns(j, h0)) {
        const float4 b = make_float4(b4[j].x + bsum, b4[j].y + bsum, b4[j].z + bsum, b4[j].w + bsum);
        if (TIED) {
          bown[j] = b;
          const uint4 fs = kHoistTables ? bfs[j] : *reinterpret_cast<const uint4 *>(p.tied_fs + h0);
          *reinterpret_cast<float4 *>(A0 + h0) =
              make_float4(b.x * lds_at(PBnext, fs.x & 0xffffu), b.y * lds_at(PBnext, fs.y & 0xffffu),
                          b.z * lds_at(PBnext, fs.z & 0xffffu), b.w * lds_at(PBnext, fs.w & 0xffffu));
        } else {
          *reinterpret_cast<float4 *>(A0 + h0) = b;
        }
        if (ALPHA_LDS) *reinterpret_cast<float4 *>(AL + h0) = areg[j];
      }
    }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      ycur[v] = ynext[v];
      if (!TIED && i0 < Ps)
        *reinterpret_cast<float4 *>(PB + i0) = make_float4(exp_limited(ycur[v].x), exp_limited(ycur[v].y),
                                                           exp_limited(ycur[v].z), exp_limited(ycur[v].w));
    }
    if (TIED) {
      float *tmp = PBcur;
      PBcur = PBnext;
      PBnext = tmp;
    }
  }
}

template <int JV, int PV, bool TIED>
static int launch_jpt(const DenParams &p, int accumulate, size_t lds_bytes, hipStream_t stream) {
  const bool want = p.deriv != nullptr;
  const bool al = p.L.alpha_in_lds;
  void (*k)(const DenParams) = nullptr;
  if (!want)
    k = den_fwd_bwd_kernel<JV, PV, true, false, false, TIED>;
  else if (accumulate)
    k = al ? den_fwd_bwd_kernel<JV, PV, true, true, true, TIED> : den_fwd_bwd_kernel<JV, PV, false, true, true, TIED>;
  else
    k = al ? den_fwd_bwd_kernel<JV, PV, true, false, true, TIED> : den_fwd_bwd_kernel<JV, PV, false, false, true, TIED>;
  TC_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds_bytes, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// accumulate != 0 selects Kaldi's "deriv += deriv_weight * gamma" form
int launch_den_mode(const DenParams &p, int accumulate, hipStream_t stream) {
  if (p.big.in_begin) return launch_den_big(p, accumulate, stream);  // graph beyond the on-chip layout
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (lds > (size_t)kLdsLimitBytes) return TC_ERR_UNSUPPORTED;
  const int JV = p.L.JV, PV = p.L.PV;
  if (p.tied_fs != nullptr) return launch_den_tied(p, accumulate, stream);  // den_tied_kernel.hip
#define TC_DISPATCH(J, V) \
  if (JV == J && PV == V) return launch_jpt<J, V, false>(p, accumulate, lds, stream);
  TC_DISPATCH(kJvSmall, kPvSmall)
  TC_DISPATCH(kJvSmall, kPvMid)
  TC_DISPATCH(kJvSmall, kPvLarge)
  TC_DISPATCH(kJvLarge, kPvSmall)
  TC_DISPATCH(kJvLarge, kPvMid)
  TC_DISPATCH(kJvLarge, kPvLarge)
#undef TC_DISPATCH
  return TC_ERR_UNSUPPORTED;
}

int launch_den(const DenParams &p, hipStream_t stream) { return launch_den_mode(p, 0, stream); }

}  // namespace tc
