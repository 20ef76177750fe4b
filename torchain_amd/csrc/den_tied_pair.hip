// Tied graphs, batches above half the chip: TWO sequences per workgroup, forward and backward recursion of a
// pair of sequences on two CUs that meet in the middle.
//
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), reached by the
// reference through src/my_lib_chain.cpp:129-131 -- the arithmetic of den_tied_kernel.hip (same factorisation
// for tied graphs, same owner-computes schedules, same fixed-point gamma).
//
// Why: the arc walk of den_tied_kernel.hip is bound by LDS gather instructions and by the VALU work of unpacking
// the cell stream (profiles/r02_*): one ds_read_b32 + one FMA per arc and sequence.  A ds_read_b64 costs the LDS
// the same two cycles as a ds_read_b32 (MI355X_MICROARCH.md, LDS table), so with the gather source interleaved
// [position][2 sequences] one gather, one unpacking and one row-end test serve TWO sequences
// (profiles/microbench/walk_pair.hip: 1.87x the sequence-cells per cycle).  Two sequences per workgroup would
// leave half the CUs idle, so the two recursions of a pair run on two workgroups at the same time:
//   role 0 ("forward")   alpha recursion t = 0 .. T of both sequences
//   role 1 ("backward")  the backward recursion with normalisers of its own (den_tied_split.hip: B_t, n_t),
//                        t = T .. 0, which does not need alpha
// and gamma_t needs alpha'_t, alpha_{t+1} and beta_{t+1} together.  With M = T / 2: until the two recursions cross,
// the forward role stores alpha'_0..alpha'_M and the backward role B_T..B_M (one history buffer: row t holds
// alpha'_t for t <= M and B_t for t > M, B_M sits in row T + 1); they exchange ONE flag each (agent-scope release /
// acquire, MI355X_MICROARCH.md "inter-workgroup visibility"); after that the forward role forms gamma_t for
// t >= M from its own alpha and the stored B_{t+1}, the backward role gamma_t for t < M from its own B and the stored
// alpha'.  HBM traffic is that of the fused kernel: every history row is written once and read once.
//
// Scale of beta: the backward role's B is beta up to a factor, beta_{t+1} = c_{t+1} B_{t+1}.  [K]'s invariant
// sum_g alpha_{t+1}(g) beta_{t+1}(g) = 1 gives c_{t+1} = 1 / sum_g alpha_{t+1}(g) B_{t+1}(g): one more value in the
// frame's block reduction.  The fixed-point gamma adds need the scale before that sum exists; they use
// c^_{t+1} from the exact recurrence c_t = c_{t+1} n_t / asum_t (den_tied_split.hip) and the conversion of the
// frame's gamma row applies c / c^ (1 +- 1e-6).
//
// Pairing does not rely on dispatch order: a workgroup takes a ticket (atomic counter, zeroed before the launch),
// pair = ticket / 2, role = ticket % 2, so the partner of a running workgroup is always one that has started or is
// the next to start.  The wait for the partner's flag is bounded; on a timeout the sequence's log-prob is NaN and
// the objective fails softly as in [K].
#include "den_tied_device.h"

namespace tc {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f2 lds_f2;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ f2 lds2(uint32_t a) { return *reinterpret_cast<lds_f2 *>(a); }

// Row sums of a lane's own rows in registers (the LDS has no room for them: at C3 the gather source of two sequences
// alone is 64 KB): one 16-element vector, row k of sequence q at element 2 k + q.  All 64 lanes of a wave are at the
// same row index, so a commit is a register write at a scalar index -- LLVM lowers the dynamic insert into a
// 16-element vector to s_set_gpr_idx_on / v_mov_b32 / s_set_gpr_idx_off (an 8-element vector would be expanded into
// eight compares and eight v_cndmask).
typedef float Rows __attribute__((ext_vector_type(16)));

struct PairCommit {
  int k;             // next row of the wave's stream (uniform)
  int K;             // own rows per lane; rows beyond are the secondary rows of hub states
  uint32_t sec_row;  // byte address of lane 0's slot of the next secondary row, sequence 0 (sequence 1: + 256)
};

__device__ __forceinline__ void addtid_st(uint32_t row, float v) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(v), "s"(row) : "memory", "m0");
}

// ADD: the row registers arrive preloaded (backward role: the self-loop arc's term of every owned state) and a
// commit adds to them -- what the thread would otherwise hold in 16 more registers across the walk.
template <bool ADD>
__device__ __forceinline__ void commit2(PairCommit &rc, Rows &r, float v0, float v1) {
  if (rc.k < rc.K) {
    if (ADD) {
      r[2 * rc.k] += v0;
      r[2 * rc.k + 1] += v1;
    } else {
      r[2 * rc.k] = v0;
      r[2 * rc.k + 1] = v1;
    }
  } else {
    addtid_st(rc.sec_row, v0);
    addtid_st(rc.sec_row + 256u, v1);
    rc.sec_row += 512u;
  }
  ++rc.k;
}

// acc(row) += w * SRC[off] for both sequences over one chunk (den_tied_device.h: do_chunk); the stream's 16-bit
// offsets are position * 8 here (cells_pair).
template <uint32_t SRC, int HALF, bool ADD>
__device__ __forceinline__ void do_chunk2(const Chunk6 &q, uint32_t m, float &acc0, float &acc1, PairCommit &rc, Rows &r) {
  // four cells at a time: a gathered cell is two registers here, and 16 waves x 4 gathers in flight keep the LDS busy
  const uint32_t oc[4] = {q.oc.x, q.oc.y, q.oc.z, q.oc.w};
  const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t o[4];
    o[0] = lo16(oc[2 * h]);
    o[1] = hi16(oc[2 * h]);
    o[2] = lo16(oc[2 * h + 1]);
    o[3] = hi16(oc[2 * h + 1]);
    f2 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = lds2(SRC + o[i]);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int i = 4 * h + i4;
      // (asm: left to the compiler the two FMAs become one v_pk_fma_f32 with the weight duplicated into a register
      // pair, and for resident chunks that duplication is hoisted out of the frame loop: +1 register per cell)
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc0) : "v"(a[i4].x), "v"(w[i]));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc1) : "v"(a[i4].y), "v"(w[i]));
      const int bit = (i & 1) ? 4 * HALF + i / 2 : 8 + 4 * HALF + i / 2;
      if (__builtin_expect((m >> bit) & 1u, 0)) {
        commit2<ADD>(rc, r, acc0, acc1);
        acc0 = 0.f;
        acc1 = 0.f;
      }
    }
  }
}

// One walk of a wave's stream for two sequences (den_tied_device.h: walk): RES resident chunks, the rest through the
// two register buffers qa / qb, which arrive PRELOADED with chunks RES and RES + 1 -- requested before the frame's
// barrier, i.e. before the frame's HBM loads (y and history rows).  A wave's vector-memory operations complete in
// order, so a chunk requested behind those loads waits for them (~2-3 k cycles under load); with the two
// preloaded chunks and the resident ones in front, that wait is over before the first such chunk is needed.  Every
// look-ahead load is unconditional (the streams end with four chunks of readable padding).
template <uint32_t SRC, int RES, bool ADD>
__device__ __forceinline__ void walk2(const Chunk6 *res, Chunk6 &qa, Chunk6 &qb, rsrc_t sbase, uint32_t lane16, int nchunks,
                                      const uint32_t *masks, PairCommit rc, Rows &r) {
  static_assert(RES % 2 == 0, "a mask word covers two chunks");
  typedef __attribute__((address_space(4))) const uint32_t const_u32;
  const_u32 *mk = (const_u32 *)masks;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < RES / 2; ++i) {
    const uint32_t m = mk[i];
    do_chunk2<SRC, 0, ADD>(res[2 * i], m, acc0, acc1, rc, r);
    do_chunk2<SRC, 1, ADD>(res[2 * i + 1], m, acc0, acc1, rc, r);
  }
  int c = RES;
  for (; c + 1 < nchunks; c += 2) {
    const uint32_t m = mk[c >> 1];
    do_chunk2<SRC, 0, ADD>(qa, m, acc0, acc1, rc, r);
    load_chunk(qa, sbase, lane16, c + 2);
    do_chunk2<SRC, 1, ADD>(qb, m, acc0, acc1, rc, r);
    load_chunk(qb, sbase, lane16, c + 3);
  }
  if (c < nchunks) do_chunk2<SRC, 0, ADD>(qa, mk[c >> 1], acc0, acc1, rc, r);
}

// N block sums behind one barrier; `red` holds N x kWaves floats and is not written again before the next barrier
template <int N>
__device__ __forceinline__ void block_sums(float (&v)[N], uint32_t red, int wave, uint32_t lane) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) ldsf_st(red + 4u * (uint32_t)(i * kWaves + wave), v[i]);
  }
  __syncthreads();
  static_assert(kWaves == 16, "one DPP row holds the wave totals");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = ldsf(red + 4u * (uint32_t)(i * kWaves) + 4u * (lane & 15u));
    t = dpp_add<0xB1>(t);
    t = dpp_add<0x4E>(t);
    t = dpp_add<0x124>(t);
    v[i] = dpp_add<0x128>(t);
  }
}

// An opaque copy of a per-thread offset, made INSIDE a loop body: every address formed from it (offset + region base +
// constant) is then formed inside the loop as well.  Formed from the loop-invariant original, the compiler hoists each
// such sum into a register of its own -- a dozen of them in these kernels -- and, short of registers, spills them;
// their reloads wait behind every store and HBM request in flight (den_tied_kernel.hip met the same with its tail).
__device__ __forceinline__ uint32_t opq(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}

// the 4 consecutive positions a thread owns in one plane, both sequences, as they lie in LDS: {p0s0, p0s1, p1s0, p1s1},
// {p2s0, p2s1, p3s0, p3s1}
struct Own8 {
  f4 lo, hi;
};
__device__ __forceinline__ Own8 own_ld(uint32_t a) { return Own8{lds4(a), lds4(a + 16u)}; }
__device__ __forceinline__ void own_st(uint32_t a, f4 s0, f4 s1) {
  lds4_st(a, f4{s0.x, s1.x, s0.y, s1.y});
  lds4_st(a + 16u, f4{s0.z, s1.z, s0.w, s1.w});
}
__device__ __forceinline__ f4 seq0(const Own8 &o) { return f4{o.lo.x, o.lo.z, o.hi.x, o.hi.z}; }
__device__ __forceinline__ f4 seq1(const Own8 &o) { return f4{o.lo.y, o.lo.w, o.hi.y, o.hi.w}; }

__device__ __forceinline__ float vload_f32(const float *ptr) {
  // a VECTOR load (never the scalar cache: the word may have been written by the partner workgroup during this launch)
  const rsrc_t r = make_rsrc(ptr, 4u);
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0)));
}

// Diagnostic build (-DTC_PAIR_STAMPS; never timed): lane 0 of every wave of pair 0 writes raw cycle stamps.
#ifdef TC_PAIR_STAMPS
#define TC_PSTAMP(role, t, i)                                                                                   \
  if (pair == 0 && lane == 0) p.pair_stamps[(((role) * kWaves + wave) * (int64_t)(T + 2) + (t)) * 8 + (i)] = clock64();
#else
#define TC_PSTAMP(role, t, i)
#endif

struct PairParams {
  uint32_t *sync;     // [0]: ticket counter, [4 + 2 * pair + role]: "my first phase is stored" flags; zeroed before the launch
  const void *fwd_cells, *bwd_cells;  // the cell streams with offsets = position * 8
  int M;              // gamma_t: t >= M by the forward role, t < M by the backward role; 1 <= M <= T - 1
  int npairs;
  int norm_stride;    // floats per sequence in fwd_norm / bwd_norm (a multiple of 32: no 128-byte line is shared)
  uint32_t aGM, aSEC, aRed;  // LDS byte offsets: gamma [2][Ps], secondary-row slots, reduction scratch
  uint32_t aY2;              // per-thread {sum y^2 of sequence 0, of sequence 1} (forward role: two registers less)
};

// reduction scratch, byte offsets from PairParams::aRed (which is 16-byte aligned)
constexpr uint32_t kScrDot = 512u, kScrFinal = 640u, kScrTicket = 896u, kScrAwait = 900u, kScrDouble = 1024u, kScrBytes = 1536u;

constexpr uint32_t kSpinSleep = 16;            // s_sleep units (64 cycles each) between two polls
constexpr uint32_t kSpinLimit = 8u << 20;      // ~ 8 M polls x ~1 k cycles: seconds

// "everything this workgroup stored so far may be read by the partner": every wave drains its stores, one lane
// releases at agent scope and raises the flag (MI355X_MICROARCH.md, valid forms: plain stores -> vmcnt(0) ->
// barrier -> release -> vmcnt(0) -> relaxed agent flag store).
__device__ __forceinline__ void publish(uint32_t *flag, uint32_t tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((gu32 *)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// one lane polls (relaxed), then ONE agent-scope acquire, its wait, the barrier; returns false on a timeout
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

constexpr uint32_t kPlane2 = 32u * kThreads;  // bytes between a thread's 4-position groups of consecutive planes, [pos][2]

// exp(y) of the thread's 4 pdfs of plane v, both sequences, into the [pdf][2] buffer
__device__ __forceinline__ void put_exp2(uint32_t base, uint32_t own32, int v, f4 y0, f4 y1) {
  own_st(base + own32 + (uint32_t)v * kPlane2, exp4(y0), exp4(y1));
}

// ---------------------------------------------------------------------------------------------------------
// The forward role.
// ---------------------------------------------------------------------------------------------------------
template <int PV, bool ACCUM, int RES1, int RES2>
__device__ __forceinline__ void pair_forward(const DenParams &p, const PairParams &q, int pair) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.S, T = p.T, M = q.M;
  const int s0 = 2 * pair;
  const bool valid1 = s0 + 1 < S;
  const int s1 = valid1 ? s0 + 1 : s0;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);
  const int K = Hs / kThreads;
  const uint32_t own16 = 16u * tid, own32 = 32u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u, kA0 = PV * 32u * kThreads;
  const uint32_t aGM0 = q.aGM, aGM1 = q.aGM + 4u * (uint32_t)Ps, aRed = q.aRed;
  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4);
  const uint32_t rb0 = 4u * (uint32_t)p.P, rb1 = valid1 ? rb0 : 0u;   // row bytes of y / deriv
  const uint32_t hb0 = 4u * (uint32_t)Hs, hb1 = valid1 ? hb0 : 0u;    // row bytes of the history
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const float leaky = p.leaky;
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist0 = p.alpha_hist + (int64_t)s0 * Hs, *const hist1 = p.alpha_hist + (int64_t)s1 * Hs;
  // (an odd batch's phantom second sequence keeps its normalisers in the spare row S: no per-lane test per frame)
  float *const fn0 = p.fwd_norm + (int64_t)s0 * q.norm_stride, *const fn1 = p.fwd_norm + (int64_t)(valid1 ? s1 : S) * q.norm_stride;
  const float *const bn0 = p.bwd_norm + (int64_t)s0 * q.norm_stride, *const bn1 = p.bwd_norm + (int64_t)(valid1 ? s1 : S) * q.norm_stride;
  auto yrow = [&](int t, int s, uint32_t bytes) __attribute__((always_inline)) { return make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, bytes); };
  auto drow = [&](int t, int s, uint32_t bytes) __attribute__((always_inline)) { return make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, bytes); };

  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky * pi * sum(pi) for both sequences
  f4 v0[2], v1[2];  // alpha_t (un-dashed) of the owned states, sequence 0 / 1
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    v0[j] = bld4(r_pi, own16, j * kPlane);
    part += hsum(v0[j]);
  }
  float asum_a = block_sum_a(part, aRed, wave, lane), asum_b = asum_a;
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (j < planes) {
      const f4 a = v0[j] + (leaky * v0[j]) * asum_a;
      own_st(kA0 + own32 + j * kPlane2, a, a);
      bst4(make_rsrc(hist0, hb0), own16 + j * kPlane, a);
      bst4(make_rsrc(hist1, hb1), own16 + j * kPlane, a);
    }
  const uint32_t aY2 = q.aY2 + 8u * tid;  // this thread's running sums of y^2 (thread-private LDS: not two registers)
  auto add_y2 = [&](float a, float b) __attribute__((always_inline)) {
    const f2 old2 = lds2(aY2);
    *reinterpret_cast<lds_f2 *>(aY2) = f2{old2.x + a, old2.y + b};
  };
  *reinterpret_cast<lds_f2 *>(aY2) = f2{0.f, 0.f};
  {
    const rsrc_t ya = yrow(0, s0, rb0), yb = yrow(0, s1, rb1);
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      if (4 * ((int)tid + kThreads * v) < Ps) {
        const f4 yp0 = row_ld(ya, own16 + v * kPlane, p.y_vec), yp1 = row_ld(yb, own16 + v * kPlane, p.y_vec);
        add_y2(hsum(yp0 * yp0), hsum(yp1 * yp1));
        put_exp2(kPB, own32, v, yp0, yp1);
        lds4_st(aGM0 + own16 + v * kPlane, mk4(0.f));
        lds4_st(aGM1 + own16 + v * kPlane, mk4(0.f));
      }
    }
  }
  if (tid == 0) {
    fn0[0] = asum_a;
    fn1[0] = asum_b;
  }
  float inv_a = __builtin_amdgcn_rcpf(asum_a), inv_b = inv_a;

  const int2 frange = p.fwd.wave_range[wave];
  const int fnch = __builtin_amdgcn_readfirstlane(frange.y) / kChunk;
  const rsrc_t fbase = make_rsrc(reinterpret_cast<const char *>(q.fwd_cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(fnch + 4) * (3 * 64 * 16));
  const uint32_t *const fmask = p.fwd.masks + wave * p.fwd.mask_stride;
  const PairCommit frc{0, K, q.aSEC + 512u * (uint32_t)p.fwd.extra_first[wave]};
  constexpr int RESMAX = RES1 > RES2 ? RES1 : RES2;
  Chunk6 fres[RESMAX > 0 ? RESMAX : 1];
#pragma unroll
  for (int i = 0; i < RESMAX; ++i) load_chunk(fres[i], fbase, lane16, i);
  float part_tot_a = 0.f, part_tot_b = 0.f;
  float c_a = 0.f, c_b = 0.f;        // exact scale c_{t-1} of the backward role's B_{t-1} (second phase)
  float chat_a = 0.f, chat_b = 0.f;  // c^_t used by the fixed-point adds of the running frame

  // secondary rows of hub states: add the slots other lanes of this wave filled to the owner's row sums
  auto fold = [&](Rows &r) __attribute__((always_inline)) {
    if (p.fwd.nfix == 0) return;
    // (buffer addressing: a flat load keeps a 64-bit per-lane pointer per table alive through the whole role)
    const rsrc_t r_fb = make_rsrc(p.fwd.fix_begin, 4u * (kThreads + 1)), r_fx = make_rsrc(p.fwd.fix, 8u * (uint32_t)p.fwd.nfix);
    const uint32_t t4 = opq(4u * tid);
    const int e0 = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fb, (int)t4, 0, 0);
    const int e1 = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fb, (int)t4, 4, 0);
    for (int e = e0; e < e1; ++e) {
      int2 f;
      f.x = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fx, 8 * e, 0, 0);
      f.y = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fx, 8 * e, 4, 0);
      const int k = 4 * (f.x / (4 * kThreads)) + (f.x & 3);
      const uint32_t src = q.aSEC + 512u * (uint32_t)((f.y - Hs - 4) >> 6) + 4u * (uint32_t)((f.y - Hs - 4) & 63);
      const float x0 = ldsf(src), x1 = ldsf(src + 256u);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        r[2 * kk] += k == kk ? x0 : 0.f;
        r[2 * kk + 1] += k == kk ? x1 : 0.f;
      }
    }
  };

  Chunk6 q0, q1;  // the stream's first two non-resident chunks
  // HBM rows a frame needs (y_t; second phase: B_t) are requested in the frame BEFORE, behind its per-state pass: a
  // wave's memory operations complete in order, so a request under the walk would hold up every streamed chunk
  // behind it for an HBM round trip; requested there, the rows arrive during the reduction and the tail.
  f4 yr0[PV], yr1[PV];  // y_t of the thread's pdfs
  f4 bt0[2] = {mk4(0.f), mk4(0.f)}, bt1[2] = {mk4(0.f), mk4(0.f)};  // B_t of the owned states
  auto request_y = [&](int t) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16);
    const rsrc_t ya = yrow(t, s0, rb0), yb = yrow(t, s1, rb1);
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      yr0[v] = row_ld(ya, o16 + v * kPlane, p.y_vec);
      yr1[v] = row_ld(yb, o16 + v * kPlane, p.y_vec);
    }
  };
  auto request_b = [&](int t) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16);
    const rsrc_t ba = make_rsrc(hist0 + (int64_t)t * hist_step, hb0), bb = make_rsrc(hist1 + (int64_t)t * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bt0[j] = bld4(ba, o16, j * kPlane);
      bt1[j] = bld4(bb, o16, j * kPlane);
    }
  };
  // One frame t: alpha_t from alpha'_{t-1}; GAMMA: also gamma_{t-1} and its derivative row from B_t.
  auto frame = [&](int t, auto res_tag, auto gamma_tag) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16), o32 = opq(own32);
    constexpr int RES = decltype(res_tag)::value;
    constexpr bool GAMMA = decltype(gamma_tag)::value;
    TC_PSTAMP(0, t, 0)
    __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready; gamma zero
    TC_PSTAMP(0, t, 1)
    float n_a = 1.f, n_b = 1.f;
    if (GAMMA && t < T) {  // n_t: c^_{t+1} = c_t asum_t / n_t, for the next frame
      n_a = vload_f32(bn0 + t);
      n_b = vload_f32(bn1 + t);
    }
    Rows r;
    age_prio_on(wave);
    walk2<kA0, RES, false>(fres, q0, q1, fbase, lane16, fnch, fmask, frc, r);
    __builtin_amdgcn_s_setprio(0);
    TC_PSTAMP(0, t, 2)
    // Behind the walk: the per-state tables (L2), then y_{t-1}.
    u4 fs[2];
    f4 ws[2], cpi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fs[j] = bld4u(r_fs, o16, j * kPlane);
      ws[j] = bld4(r_ws, o16, j * kPlane);
    }
    f4 yp0[PV], yp1[PV];
    fold(r);
    float sums[GAMMA ? 4 : 2];
    constexpr int GA = GAMMA ? 2 : 0, GB = GAMMA ? 3 : 1;  // (the gamma frames' extra sums; in range either way)
#pragma unroll
    for (int i = 0; i < (GAMMA ? 4 : 2); ++i) sums[i] = 0.f;
    const float gsa = kGammaScale * chat_a, gsb = kGammaScale * chat_b;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      v0[j] = v1[j] = mk4(0.f);
      if (j < planes) {
        const Own8 al = own_ld(kA0 + o32 + j * kPlane2);  // alpha'_{t-1} of the owned states
        const f4 al0 = seq0(al), al1 = seq1(al);
        // alpha_t(g) asum_{t-1} = p(f(g)) F(g) + p(s(g)) w_s alpha'_{t-1}(g); with B_t(g): the two parts are the
        // occupations of the forward-class arcs into g and of its self-loop in frame t-1
        auto one = [&](uint32_t fsx, float wsx, float F0, float F1, float a0, float a1, float b0, float b1, float &o0, float &o1) __attribute__((always_inline)) {
          const f2 pf = lds2(kPB + 2u * (fsx & 0xffffu)), ps = lds2(kPB + 2u * (fsx >> 16));
          const float fp0 = pf.x * F0 * inv_a, fp1 = pf.y * F1 * inv_b;
          const float sp0 = ps.x * (wsx * a0) * inv_a, sp1 = ps.y * (wsx * a1) * inv_b;
          o0 = fp0 + sp0;
          o1 = fp1 + sp1;
          if constexpr (GAMMA) {
            const float g0 = gsa * b0, g1 = gsb * b1;
            gamma_add_a(aGM0 + (fsx & 0xffffu), g0 * fp0);
            gamma_add_a(aGM0 + (fsx >> 16), g0 * sp0);
            gamma_add_a(aGM1 + (fsx & 0xffffu), g1 * fp1);
            gamma_add_a(aGM1 + (fsx >> 16), g1 * sp1);
            sums[GA] = fmaf(o0, b0, sums[GA]);
            sums[GB] = fmaf(o1, b1, sums[GB]);
          }
        };
        float oa[4], ob[4];
        one(fs[j].x, ws[j].x, r[8 * j + 2 * 0], r[8 * j + 2 * 0 + 1], al0.x, al1.x, bt0[j].x, bt1[j].x, oa[0], ob[0]);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].y, ws[j].y, r[8 * j + 2 * 1], r[8 * j + 2 * 1 + 1], al0.y, al1.y, bt0[j].y, bt1[j].y, oa[1], ob[1]);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].z, ws[j].z, r[8 * j + 2 * 2], r[8 * j + 2 * 2 + 1], al0.z, al1.z, bt0[j].z, bt1[j].z, oa[2], ob[2]);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].w, ws[j].w, r[8 * j + 2 * 3], r[8 * j + 2 * 3 + 1], al0.w, al1.w, bt0[j].w, bt1[j].w, oa[3], ob[3]);
        __builtin_amdgcn_sched_barrier(0);
        v0[j] = f4{oa[0], oa[1], oa[2], oa[3]};
        v1[j] = f4{ob[0], ob[1], ob[2], ob[3]};
        sums[0] += hsum(v0[j]);
        sums[1] += hsum(v1[j]);
      }
    }
    // pi: first touched behind the reduction, which hides its L2 trip
#pragma unroll
    for (int j = 0; j < 2; ++j) cpi[j] = bld4(r_pi, o16, j * kPlane);
    if (GAMMA) {
      // y_{t-1} (the derivative row's l2 term) is read again rather than held in registers across the frame: the row
      // was this CU's a frame ago (L2), and its first use is behind the reduction
      const rsrc_t ya = yrow(t - 1, s0, rb0), yb = yrow(t - 1, s1, rb1);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        yp0[v] = row_ld(ya, o16 + v * kPlane, p.y_vec);
        yp1[v] = row_ld(yb, o16 + v * kPlane, p.y_vec);
      }
      // the next frame's B row (unconditional, index clamped: a conditional assignment would keep the old values alive
      // through the whole frame)
      request_b(t + 1 <= T ? t + 1 : T);
    }
    // the history row of frame t-1 (first phase), behind everything this frame still loads
    if (!GAMMA && t > 1) {
      const rsrc_t ha = make_rsrc(hist0 + (int64_t)(t - 1) * hist_step, hb0), hb = make_rsrc(hist1 + (int64_t)(t - 1) * hist_step, hb1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j < planes) {
          const Own8 o = own_ld(kA0 + o32 + j * kPlane2);  // alpha'_{t-1} of the owned states: still in the gather buffer
          bst4(ha, o16 + j * kPlane, seq0(o));
          bst4(hb, o16 + j * kPlane, seq1(o));
        }
    }
    TC_PSTAMP(0, t, 3)
    block_sums(sums, aRed, wave, lane);  // its barrier also ends every wave's gathers and completes gamma_{t-1}
    TC_PSTAMP(0, t, 4)
    asum_a = sums[0];
    asum_b = sums[1];
    part_tot_a = part_tot_b = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        const f4 a0 = v0[j] + (leaky * cpi[j]) * asum_a, a1 = v1[j] + (leaky * cpi[j]) * asum_b;
        own_st(kA0 + o32 + j * kPlane2, a0, a1);
        part_tot_a += hsum(a0);
        part_tot_b += hsum(a1);
      }
    // the next frame's first chunks: requested ahead of the stores and of the HBM request below (in-order completion)
    load_chunk(q0, fbase, lane16, RES);
    load_chunk(q1, fbase, lane16, RES + 1);
    if (GAMMA) {
      // the derivative row of frame t-1: gamma_{t-1} * (c_t / c^_t)
      const float ca = __builtin_amdgcn_rcpf(sums[GA]), cb = __builtin_amdgcn_rcpf(sums[GB]);
      const float sa = kGammaInvScale * (ca * __builtin_amdgcn_rcpf(chat_a)), sb = kGammaInvScale * (cb * __builtin_amdgcn_rcpf(chat_b));
      const rsrc_t da = drow(t - 1, s0, rb0), db = drow(t - 1, s1, rb1);
#pragma unroll
      for (int v = 0; v < PV; ++v)
        if (4 * ((int)tid + kThreads * v) < Ps) {
          const u4 ga = lds4u(aGM0 + o16 + v * kPlane), gb = lds4u(aGM1 + o16 + v * kPlane);
          lds4_st(aGM0 + o16 + v * kPlane, mk4(0.f));
          lds4_st(aGM1 + o16 + v * kPlane, mk4(0.f));
          f4 oa = (p.deriv_weight * sa) * f4{(float)ga.x, (float)ga.y, (float)ga.z, (float)ga.w} - p.l2_scale * yp0[v];
          f4 ob = (p.deriv_weight * sb) * f4{(float)gb.x, (float)gb.y, (float)gb.z, (float)gb.w} - p.l2_scale * yp1[v];
          if (ACCUM) {
            oa += row_ld(da, o16 + v * kPlane, p.d_vec);
            ob += row_ld(db, o16 + v * kPlane, p.d_vec);
          }
          row_st(da, o16 + v * kPlane, p.d_vec, oa);
          row_st(db, o16 + v * kPlane, p.d_vec, ob);
        }
      // c^_{t+1} = c_t asum_t / n_t
      chat_a = ca * asum_a * __builtin_amdgcn_rcpf(n_a);
      chat_b = cb * asum_b * __builtin_amdgcn_rcpf(n_b);
      c_a = ca;
      c_b = cb;
    }
    if (t < T) {
#pragma unroll
      for (int v = 0; v < PV; ++v)
        if (4 * ((int)tid + kThreads * v) < Ps) {
          add_y2(hsum(yr0[v] * yr0[v]), hsum(yr1[v] * yr1[v]));
          put_exp2(kPB, o32, v, yr0[v], yr1[v]);
        }
    }
    if (tid == 0) {
      fn0[t] = asum_a;
      fn1[t] = asum_b;
    }
    inv_a = __builtin_amdgcn_rcpf(asum_a);
    inv_b = __builtin_amdgcn_rcpf(asum_b);
    request_y(t + 1 < T ? t + 1 : T - 1);  // the next frame's y row (HBM): its first use is that frame's tail
    TC_PSTAMP(0, t, 5)
  };

  // ---- first phase: frames 1 .. M, pure recursion, rows 0 .. M-1 stored under the walks
  load_chunk(q0, fbase, lane16, RES1);
  load_chunk(q1, fbase, lane16, RES1 + 1);
  request_y(1);  // (T >= 2)
#ifndef TC_PAIR_NO_PURE
  for (int t = 1; t <= M; ++t) frame(t, std::integral_constant<int, RES1>(), std::false_type());
#endif
  {
    // row M (still in the gather buffer), then the hand-off
    const rsrc_t ha = make_rsrc(hist0 + (int64_t)M * hist_step, hb0), hb = make_rsrc(hist1 + (int64_t)M * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        const Own8 o = own_ld(kA0 + own32 + j * kPlane2);
        bst4(ha, own16 + j * kPlane, seq0(o));
        bst4(hb, own16 + j * kPlane, seq1(o));
      }
  }
  TC_PSTAMP(0, T + 1, 0)
  publish(q.sync + 4 + 2 * pair, tid);
  const bool partner_ok = await(q.sync + 4 + 2 * pair + 1, tid, aRed + kScrAwait);
  TC_PSTAMP(0, T + 1, 1)
  {
    // c_M = 1 / sum_g alpha_M(g) B_M(g)  (B_M: row T + 1), c^_{M+1} = c_M asum_M / n_M
    const rsrc_t ba = make_rsrc(hist0 + (int64_t)(T + 1) * hist_step, hb0), bb = make_rsrc(hist1 + (int64_t)(T + 1) * hist_step, hb1);
    float d[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        d[0] += hsum(v0[j] * bld4(ba, own16, j * kPlane));
        d[1] += hsum(v1[j] * bld4(bb, own16, j * kPlane));
      }
    block_sums(d, aRed + kScrDot, wave, lane);
    c_a = __builtin_amdgcn_rcpf(d[0]);
    c_b = __builtin_amdgcn_rcpf(d[1]);
    chat_a = c_a * asum_a * __builtin_amdgcn_rcpf(vload_f32(bn0 + M));
    chat_b = c_b * asum_b * __builtin_amdgcn_rcpf(vload_f32(bn1 + M));
  }
  // ---- second phase: frames M+1 .. T with gamma_{t-1}
  load_chunk(q0, fbase, lane16, RES2);
  load_chunk(q1, fbase, lane16, RES2 + 1);
  request_b(M + 1);
#ifndef TC_PAIR_NO_GAMMA
  for (int t = M + 1; t <= T; ++t) frame(t, std::integral_constant<int, RES2>(), std::true_type());
#endif

  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h); log-prob = log tot + sum_{t<T} log asum_t
  const f2 y2 = lds2(aY2);
  float fin[4] = {part_tot_a, part_tot_b, y2.x, y2.y};
  block_sums(fin, aRed + kScrFinal, wave, lane);
  {
    // the asum_t this workgroup wrote (same CU: its own L1 / L2 path), summed as (double) logf like the fused kernel
    double la = 0.0, lb = 0.0;
    const rsrc_t na = make_rsrc(fn0, 4u * (uint32_t)T), nb = make_rsrc(fn1, 4u * (uint32_t)T);
    for (int u = (int)tid; u < T; u += kThreads) {
      la += (double)__logf(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(na, 4 * u, 0, 0)));
      lb += (double)__logf(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(nb, 4 * u, 0, 0)));
    }
    for (int off = 32; off > 0; off >>= 1) {
      la += __shfl_xor(la, off);
      lb += __shfl_xor(lb, off);
    }
    typedef __attribute__((address_space(3))) double lds_d;
    const uint32_t aD = aRed + kScrDouble;
    if (lane == 0) {
      *reinterpret_cast<lds_d *>(aD + 16u * (uint32_t)wave) = la;
      *reinterpret_cast<lds_d *>(aD + 16u * (uint32_t)wave + 8u) = lb;
    }
    __syncthreads();
    if (tid == 0) {
      double sa = 0.0, sb = 0.0;
      for (int w = 0; w < kWaves; ++w) {
        sa += *reinterpret_cast<lds_d *>(aD + 16u * (uint32_t)w);
        sb += *reinterpret_cast<lds_d *>(aD + 16u * (uint32_t)w + 8u);
      }
      const double bad = partner_ok ? 0.0 : (double)__builtin_nanf("");
      const double y2da = (double)fin[2], y2db = (double)fin[3];
      p.seq_logprob[s0] = sa + (double)__logf(fin[0]) + (y2da - y2da) + bad;  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s0] = y2da;
      if (valid1) {
        p.seq_logprob[s1] = sb + (double)__logf(fin[1]) + (y2db - y2db) + bad;
        p.seq_y2[s1] = y2db;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// The backward role: den_tied_split.hip's recursion B'_T = 1, U_t(h) = sum_out w B_{t+1}(g) p_t(f(g)) + p_t(s(h)) w_s(h) B_{t+1}(h),
// n_t = sum_h U_t(h) / H, B'_t = U_t / n_t, B_t = B'_t + leaky sum_h pi(h) B'_t(h); in its second phase also gamma_t and
// the derivative row with the fused kernel's per-state formulas (den_tied_kernel.hip) and beta_{t+1} = c_{t+1} B_{t+1}.
// LDS: one exp(y) buffer (rewritten in place behind a barrier, the fused kernel's tight layout).
//
// Every frame is the same walk + per-state pass: the self-loop arc's term of U_t waits in the row registers the walk
// adds to (form_y), so B_{t+1} is not held across the walk.  In the second phase the per-state gamma work of frame t
// (fixed-point adds from B_{t+1}, alpha'_t, alpha'_{t+1} and exp(y_t)) is done at the END of frame t + 1, between
// forming B_{t+1} and the next walk, where nothing of a walk or a pass is in registers; frame t then only converts
// the finished gamma_t row behind its reduction.
// ---------------------------------------------------------------------------------------------------------
template <int PV, bool ACCUM, int RES1, int RES2>
__device__ __forceinline__ void pair_backward(const DenParams &p, const PairParams &q, int pair) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.S, T = p.T, M = q.M, H = p.H;
  const int s0 = 2 * pair;
  const bool valid1 = s0 + 1 < S;
  const int s1 = valid1 ? s0 + 1 : s0;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);
  const int K = Hs / kThreads;
  const uint32_t own16 = 16u * tid, own32 = 32u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u, kA0 = PV * 32u * kThreads;
  const uint32_t aGM0 = q.aGM, aGM1 = q.aGM + 4u * (uint32_t)Ps, aRed = q.aRed;
  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4);
  const uint32_t rb0 = 4u * (uint32_t)p.P, rb1 = valid1 ? rb0 : 0u;
  const uint32_t hb0 = 4u * (uint32_t)Hs, hb1 = valid1 ? hb0 : 0u;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const float leaky = p.leaky;
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist0 = p.alpha_hist + (int64_t)s0 * Hs, *const hist1 = p.alpha_hist + (int64_t)s1 * Hs;
  const float *const fn0 = p.fwd_norm + (int64_t)s0 * q.norm_stride, *const fn1 = p.fwd_norm + (int64_t)(valid1 ? s1 : S) * q.norm_stride;
  float *const bn0 = p.bwd_norm + (int64_t)s0 * q.norm_stride, *const bn1 = p.bwd_norm + (int64_t)(valid1 ? s1 : S) * q.norm_stride;
  auto yrow = [&](int t, int s, uint32_t bytes) __attribute__((always_inline)) { return make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, bytes); };
  auto drow = [&](int t, int s, uint32_t bytes) __attribute__((always_inline)) { return make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, bytes); };
  const float inv_h = 1.0f / (float)H;

  const int2 brange = p.bwd.wave_range[wave];
  const int bnch = __builtin_amdgcn_readfirstlane(brange.y) / kChunk;
  const rsrc_t bbase = make_rsrc(reinterpret_cast<const char *>(q.bwd_cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(brange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(bnch + 4) * (3 * 64 * 16));
  const uint32_t *const bmask = p.bwd.masks + wave * p.bwd.mask_stride;
  const PairCommit brc{0, K, q.aSEC + 512u * (uint32_t)p.bwd.extra_first[wave]};
  constexpr int RESMAX = RES1 > RES2 ? RES1 : RES2;
  Chunk6 bres[RESMAX > 0 ? RESMAX : 1];
#pragma unroll
  for (int i = 0; i < RESMAX; ++i) load_chunk(bres[i], bbase, lane16, i);

  auto fold = [&](Rows &r) __attribute__((always_inline)) {
    if (p.bwd.nfix == 0) return;
    // (buffer addressing: a flat load keeps a 64-bit per-lane pointer per table alive through the whole role)
    const rsrc_t r_fb = make_rsrc(p.bwd.fix_begin, 4u * (kThreads + 1)), r_fx = make_rsrc(p.bwd.fix, 8u * (uint32_t)p.bwd.nfix);
    const uint32_t t4 = opq(4u * tid);
    const int e0 = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fb, (int)t4, 0, 0);
    const int e1 = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fb, (int)t4, 4, 0);
    for (int e = e0; e < e1; ++e) {
      int2 f;
      f.x = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fx, 8 * e, 0, 0);
      f.y = (int)__builtin_amdgcn_raw_buffer_load_b32(r_fx, 8 * e, 4, 0);
      const int k = 4 * (f.x / (4 * kThreads)) + (f.x & 3);
      const uint32_t src = q.aSEC + 512u * (uint32_t)((f.y - Hs - 4) >> 6) + 4u * (uint32_t)((f.y - Hs - 4) & 63);
      const float x0 = ldsf(src), x1 = ldsf(src + 256u);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        r[2 * kk] += k == kk ? x0 : 0.f;
        r[2 * kk + 1] += k == kk ? x1 : 0.f;
      }
    }
  };

  // B'_T = 1, B_T = 1 + leaky * sum(pi)
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (j < planes) part += hsum(leaky * bld4(r_pi, own16, j * kPlane));
  const float bsum_T = block_sum_a(part, aRed, wave, lane);
  f4 bo0[2], bo1[2];  // B_{t+1} of the owned states
  {
    const rsrc_t ha = make_rsrc(hist0 + (int64_t)T * hist_step, hb0), hb = make_rsrc(hist1 + (int64_t)T * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bo0[j] = bo1[j] = mk4(0.f);
      if (j < planes) {
        const int h0 = 4 * ((int)tid + kThreads * j);
        const float b = 1.0f + bsum_T;
        bo0[j] = bo1[j] = f4{h0 < H ? b : 0.f, h0 + 1 < H ? b : 0.f, h0 + 2 < H ? b : 0.f, h0 + 3 < H ? b : 0.f};
        bst4(ha, own16 + j * kPlane, bo0[j]);
        bst4(hb, own16 + j * kPlane, bo1[j]);
      }
    }
    const rsrc_t ya = yrow(T - 1, s0, rb0), yb = yrow(T - 1, s1, rb1);
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      if (4 * ((int)tid + kThreads * v) < Ps) {
        put_exp2(kPB, own32, v, row_ld(ya, own16 + v * kPlane, p.y_vec), row_ld(yb, own16 + v * kPlane, p.y_vec));
        lds4_st(aGM0 + own16 + v * kPlane, mk4(0.f));
        lds4_st(aGM1 + own16 + v * kPlane, mk4(0.f));
      }
    }
    __syncthreads();  // exp(y_{T-1}) complete
  }
  // From B_{t+1} (bo) and exp(y_t) in LDS: the next walk's gather source Y_t(g) = B_{t+1}(g) p_t(f(g)) and the self-loop
  // arc's term of U_t(g), p_t(s(g)) w_s(g) B_{t+1}(g), which waits in the row registers the walk adds to.
  Rows rnext;
  auto form_y = [&](const u4 (&fs)[2], const f4 (&ws)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rnext[8 * j + i] = 0.f;
      if (j < planes) {
        const f2 p0 = lds2(kPB + 2u * (fs[j].x & 0xffffu)), p1 = lds2(kPB + 2u * (fs[j].y & 0xffffu));
        const f2 p2 = lds2(kPB + 2u * (fs[j].z & 0xffffu)), p3 = lds2(kPB + 2u * (fs[j].w & 0xffffu));
        own_st(kA0 + own32 + j * kPlane2, bo0[j] * f4{p0.x, p1.x, p2.x, p3.x}, bo1[j] * f4{p0.y, p1.y, p2.y, p3.y});
        const f2 q0 = lds2(kPB + 2u * (fs[j].x >> 16)), q1 = lds2(kPB + 2u * (fs[j].y >> 16));
        const f2 q2 = lds2(kPB + 2u * (fs[j].z >> 16)), q3 = lds2(kPB + 2u * (fs[j].w >> 16));
        const f4 sa = bo0[j] * ws[j] * f4{q0.x, q1.x, q2.x, q3.x}, sb = bo1[j] * ws[j] * f4{q0.y, q1.y, q2.y, q3.y};
        rnext[8 * j + 0] = sa.x;
        rnext[8 * j + 1] = sb.x;
        rnext[8 * j + 2] = sa.y;
        rnext[8 * j + 3] = sb.y;
        rnext[8 * j + 4] = sa.z;
        rnext[8 * j + 5] = sb.z;
        rnext[8 * j + 6] = sa.w;
        rnext[8 * j + 7] = sb.w;
      }
    }
  };
  {
    u4 fs[2];
    f4 ws[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fs[j] = bld4u(r_fs, own16, j * kPlane);
      ws[j] = bld4(r_ws, own16, j * kPlane);
    }
    form_y(fs, ws);
  }
  Chunk6 q0, q1;  // the stream's first two non-resident chunks of the NEXT frame: requested before the frame's stores
  // HBM rows a frame needs are requested a frame ahead or behind the frame's own pass (see the forward role).
  f4 yn0[PV], yn1[PV];  // y_{t-1}
  f4 al0[2] = {mk4(0.f), mk4(0.f)}, al1[2] = {mk4(0.f), mk4(0.f)};  // alpha'_{t-1}: gamma_{t-1} is formed at the end of frame t
  f4 au0[2] = {mk4(0.f), mk4(0.f)}, au1[2] = {mk4(0.f), mk4(0.f)};  // alpha'_t
  auto request_yn = [&](int t) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16);
    const int tn = t > 0 ? t - 1 : 0;
    const rsrc_t ya = yrow(tn, s0, rb0), yb = yrow(tn, s1, rb1);
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      yn0[v] = row_ld(ya, o16 + v * kPlane, p.y_vec);
      yn1[v] = row_ld(yb, o16 + v * kPlane, p.y_vec);
    }
  };
  // alpha'_{tl} -> al (HBM: requested behind the pass), alpha'_{tl+1} -> au (this CU's a frame ago, L2: requested
  // behind the derivative row, whose conversion needs the registers)
  auto request_al = [&](int tl) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16);
    const rsrc_t a0r = make_rsrc(hist0 + (int64_t)tl * hist_step, hb0), a1r = make_rsrc(hist1 + (int64_t)tl * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      al0[j] = bld4(a0r, o16, j * kPlane);
      al1[j] = bld4(a1r, o16, j * kPlane);
    }
  };
  auto request_au = [&](int tl) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16);
    const rsrc_t u0r = make_rsrc(hist0 + (int64_t)(tl + 1) * hist_step, hb0), u1r = make_rsrc(hist1 + (int64_t)(tl + 1) * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      au0[j] = bld4(u0r, o16, j * kPlane);
      au1[j] = bld4(u1r, o16, j * kPlane);
    }
  };
  float chat_a = 0.f, chat_b = 0.f;  // c^_{t+1}: the scale of B_{t+1} the fixed-point adds of gamma_t used
  float dpart_a = 0.f, dpart_b = 0.f;  // this thread's part of sum_g alpha_{t+1}(g) B_{t+1}(g) = 1 / c_{t+1}

  // gamma_tl's fixed-point adds, from B_{tl+1} (bo), alpha'_tl (al), alpha'_{tl+1} (au) and exp(y_tl) (LDS) -- the fused
  // kernel's per-state formulas (den_tied_kernel.hip):
  //   self-loop     occ_s = beta_{tl+1}(g) selfpart,  selfpart = p_tl(s(g)) w_s(g) alpha'_tl(g) / asum_tl
  //   forward class occ_f = beta_{tl+1}(g) (alpha_{tl+1}(g) - selfpart),  alpha_{tl+1} = alpha'_{tl+1} - leaky pi asum_{tl+1}
  // with beta_{tl+1} = c^ B_{tl+1}.  Leaves this thread's part of sum_g alpha_{tl+1}(g) B_{tl+1}(g) in dpart.
  auto gamma_block = [&](int tl, const u4 (&fs)[2], const f4 (&ws)[2], const f4 (&cp)[2]) {
    const float as_a = vload_f32(fn0 + tl), as_b = vload_f32(fn1 + tl);
    const float asu_a = vload_f32(fn0 + tl + 1), asu_b = vload_f32(fn1 + tl + 1);
    const float inv_as_a = __builtin_amdgcn_rcpf(as_a), inv_as_b = __builtin_amdgcn_rcpf(as_b);
    const float gsa = kGammaScale * chat_a, gsb = kGammaScale * chat_b;
    dpart_a = dpart_b = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        auto one = [&](uint32_t fsx, float wsx, float cpx, float b0, float b1, float a0, float a1, float up0, float up1) __attribute__((always_inline)) {
          const f2 ps = lds2(kPB + 2u * (fsx >> 16));
          const float sp0 = ps.x * wsx * a0 * inv_as_a, sp1 = ps.y * wsx * a1 * inv_as_b;
          const float an0 = up0 - cpx * asu_a, an1 = up1 - cpx * asu_b;
          const float g0 = gsa * b0, g1 = gsb * b1;
          gamma_add_a(aGM0 + (fsx >> 16), g0 * sp0);
          gamma_add_a(aGM0 + (fsx & 0xffffu), g0 * fmaxf(an0 - sp0, 0.f));
          gamma_add_a(aGM1 + (fsx >> 16), g1 * sp1);
          gamma_add_a(aGM1 + (fsx & 0xffffu), g1 * fmaxf(an1 - sp1, 0.f));
          dpart_a = fmaf(an0, b0, dpart_a);
          dpart_b = fmaf(an1, b1, dpart_b);
        };
        one(fs[j].x, ws[j].x, cp[j].x, bo0[j].x, bo1[j].x, al0[j].x, al1[j].x, au0[j].x, au1[j].x);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].y, ws[j].y, cp[j].y, bo0[j].y, bo1[j].y, al0[j].y, al1[j].y, au0[j].y, au1[j].y);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].z, ws[j].z, cp[j].z, bo0[j].z, bo1[j].z, al0[j].z, al1[j].z, au0[j].z, au1[j].z);
        __builtin_amdgcn_sched_barrier(0);
        one(fs[j].w, ws[j].w, cp[j].w, bo0[j].w, bo1[j].w, al0[j].w, al1[j].w, au0[j].w, au1[j].w);
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // One frame t: B_t from B_{t+1}; GAMMA: also the derivative row of frame t (its gamma was formed at the end of frame
  // t + 1) and, at its end, gamma_{t-1}.
  // (LAST: frame 0, instantiated on its own -- an early exit inside the loop's frame would make every assignment of
  // the tail conditional and so keep last frame's values alive through the whole frame)
  auto frame = [&](int t, auto res_tag, auto gamma_tag, auto last_tag) __attribute__((always_inline)) {
    const uint32_t o16 = opq(own16), o32 = opq(own32);
    constexpr int RES = decltype(res_tag)::value;
    constexpr bool GAMMA = decltype(gamma_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    TC_PSTAMP(1, t, 0)
    __syncthreads();  // Y_t and exp(y_t) ready; gamma_t complete
    TC_PSTAMP(1, t, 1)
    Rows r = rnext;
    age_prio_on(wave);
    walk2<kA0, RES, true>(bres, q0, q1, bbase, lane16, bnch, bmask, brc, r);
    __builtin_amdgcn_s_setprio(0);
    TC_PSTAMP(1, t, 2)
    // Behind the walk: the per-state tables (L2)
    u4 fs[2];
    f4 ws[2], cp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fs[j] = bld4u(r_fs, o16, j * kPlane);
      ws[j] = bld4(r_ws, o16, j * kPlane);
      cp[j] = leaky * bld4(r_pi, o16, j * kPlane);
    }
    if (GAMMA) {
      // the frame's HBM rows, behind the tables (in-order completion: the pass waits for the tables only): alpha'_{t-1}
      // for gamma_{t-1} at the end of this frame, y_{t-1} for the exp(y) rewrite behind the derivative row
      request_al(LAST ? 0 : t - 1);
      request_yn(t);
    }
    fold(r);
    constexpr int NS = GAMMA ? 6 : 4;
    float sums[NS];
    constexpr int GA = GAMMA ? 4 : 0, GB = GAMMA ? 5 : 1;  // (the gamma frames' extra sums; in range either way)
#pragma unroll
    for (int i = 0; i < NS; ++i) sums[i] = 0.f;
    f4 u0[2], u1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // U_t(h): the self-loop arc's term was preloaded into the row registers (form_y)
      u0[j] = f4{r[8 * j + 0], r[8 * j + 2], r[8 * j + 4], r[8 * j + 6]};
      u1[j] = f4{r[8 * j + 1], r[8 * j + 3], r[8 * j + 5], r[8 * j + 7]};
      if (j < planes) {
        sums[0] += hsum(u0[j]);
        sums[1] += hsum(u1[j]);
        sums[2] += hsum(cp[j] * u0[j]);
        sums[3] += hsum(cp[j] * u1[j]);
      }
    }
    f4 yc0[PV], yc1[PV];
    if (GAMMA) {
      sums[GA] = dpart_a;
      sums[GB] = dpart_b;
      // Behind the pass, first used behind the reduction: y_t (the l2 term of the derivative row; this CU read it a
      // frame ago: L2)
      const rsrc_t ya = yrow(t, s0, rb0), yb = yrow(t, s1, rb1);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        yc0[v] = row_ld(ya, o16 + v * kPlane, p.y_vec);
        yc1[v] = row_ld(yb, o16 + v * kPlane, p.y_vec);
      }
    }
    TC_PSTAMP(1, t, 3)
    block_sums(sums, aRed, wave, lane);  // its barrier also ends every wave's gathers of Y_t
    TC_PSTAMP(1, t, 4)
    const float n_a = sums[0] * inv_h, n_b = sums[1] * inv_h;
    const float inv_n_a = __builtin_amdgcn_rcpf(n_a), inv_n_b = __builtin_amdgcn_rcpf(n_b);
    const float bsum_a = sums[2] * inv_n_a, bsum_b = sums[3] * inv_n_b;
    if (tid == 0) {
      bn0[t] = __builtin_amdgcn_rcpf(inv_n_a);  // the normaliser actually applied (its reciprocal is what B was scaled by)
      bn1[t] = __builtin_amdgcn_rcpf(inv_n_b);
    }
    if (GAMMA) {
      const float ca = __builtin_amdgcn_rcpf(sums[GA]), cb = __builtin_amdgcn_rcpf(sums[GB]);  // c_{t+1}
      const float sa = kGammaInvScale * (ca * __builtin_amdgcn_rcpf(chat_a)), sb = kGammaInvScale * (cb * __builtin_amdgcn_rcpf(chat_b));
      const rsrc_t da = drow(t, s0, rb0), db = drow(t, s1, rb1);
      float gs[2] = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < PV; ++v)
        if (4 * ((int)tid + kThreads * v) < Ps) {
          {  // (one sequence after the other: this is where the role's register use peaks)
            const u4 ga = lds4u(aGM0 + o16 + v * kPlane);
            lds4_st(aGM0 + o16 + v * kPlane, mk4(0.f));
            const f4 g0 = sa * f4{(float)ga.x, (float)ga.y, (float)ga.z, (float)ga.w};
            if (LAST) gs[0] += hsum(g0);
            f4 oa = p.deriv_weight * g0 - p.l2_scale * yc0[v];
            if (ACCUM) oa += row_ld(da, o16 + v * kPlane, p.d_vec);
            row_st(da, o16 + v * kPlane, p.d_vec, oa);
          }
          __builtin_amdgcn_sched_barrier(0);
          {
            const u4 gb = lds4u(aGM1 + o16 + v * kPlane);
            lds4_st(aGM1 + o16 + v * kPlane, mk4(0.f));
            const f4 g1 = sb * f4{(float)gb.x, (float)gb.y, (float)gb.z, (float)gb.w};
            if (LAST) gs[1] += hsum(g1);
            f4 ob = p.deriv_weight * g1 - p.l2_scale * yc1[v];
            if (ACCUM) ob += row_ld(db, o16 + v * kPlane, p.d_vec);
            row_st(db, o16 + v * kPlane, p.d_vec, ob);
          }
        }
      const float as_a = vload_f32(fn0 + t), as_b = vload_f32(fn1 + t);
      const float inv_as_a = __builtin_amdgcn_rcpf(as_a), inv_as_b = __builtin_amdgcn_rcpf(as_b);
      if (LAST) {
        // [K] BetaGeneralFrameDebug(0): alpha'_0 . beta'_0 and sum(gamma_0) must both be ~1 per sequence;
        // beta'_0 = c_0 U_0 / n_0 = c_1 U_0 / asum_0  (request_al(0) above left alpha'_0 in al)
        float fin[4] = {gs[0], gs[1], 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < planes) {
            fin[2] += hsum(al0[j] * u0[j]);
            fin[3] += hsum(al1[j] * u1[j]);
          }
        block_sums(fin, aRed + kScrFinal, wave, lane);
        if (tid == 0) {
          p.seq_ab[s0] = fin[2] * ca * inv_as_a;
          p.seq_gsum[s0] = fin[0];
          if (valid1) {
            p.seq_ab[s1] = fin[3] * cb * inv_as_b;
            p.seq_gsum[s1] = fin[1];
          }
        }
        return;
      }
      // c_t = c_{t+1} n_t / asum_t: the scale of B_t in gamma_{t-1}'s adds below
      chat_a = ca * n_a * inv_as_a;
      chat_b = cb * n_b * inv_as_b;
    }
    // exp(y_{t-1}) overwrites exp(y_t) in place -- its readers (the per-state pass) are behind the reduction's
    // barrier -- and one more barrier publishes it to the Y update below
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      if (4 * ((int)tid + kThreads * v) < Ps) put_exp2(kPB, o32, v, yn0[v], yn1[v]);
    }
    TC_PSTAMP(1, t, 5)
    __syncthreads();
    TC_PSTAMP(1, t, 6)
    // B_t = B'_t + leaky-sum, the next frame's gather source Y_{t-1} and self-loop terms; second phase: gamma_{t-1};
    // then the next frame's first chunk requests and, behind them, B_t's history row (first phase: rows T-1 .. M+1)
    // (unconditional for both planes: a conditional assignment would keep the old B alive through the whole frame)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bo0[j] = u0[j] * inv_n_a + bsum_a;
      bo1[j] = u1[j] * inv_n_b + bsum_b;
    }
    if (!GAMMA) {
      // First phase.  The tail has a memory part (the next frame's first chunks, B_t's history row, the next y row)
      // and an LDS part (form_y's gathers and stores), independent of each other, and all 16 waves reach it together
      // behind the barrier above: half of the waves take the memory part first, the other half the LDS part, so that
      // the memory pipe and the LDS work side by side instead of one after the other.
      auto memory_part = [&]() __attribute__((always_inline)) {
        load_chunk(q0, bbase, lane16, RES);  // (ahead of the stores and of the HBM request: in-order completion)
        load_chunk(q1, bbase, lane16, RES + 1);
        if (t > M) {
          const rsrc_t ha = make_rsrc(hist0 + (int64_t)t * hist_step, hb0), hb = make_rsrc(hist1 + (int64_t)t * hist_step, hb1);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j < planes) {
              bst4(ha, o16 + j * kPlane, bo0[j]);
              bst4(hb, o16 + j * kPlane, bo1[j]);
            }
        }
        request_yn(t - 1);  // the next frame's y row (HBM, unconditional: see the forward role): first used behind its pass
      };
      if (wave < kWaves / 2) {
        memory_part();
        __builtin_amdgcn_sched_barrier(0);
        form_y(fs, ws);
      } else {
        form_y(fs, ws);
        __builtin_amdgcn_sched_barrier(0);
        memory_part();
      }
    } else {
      // (the gamma frames have no registers for the chunks during gamma_block)
      form_y(fs, ws);
      gamma_block(t - 1, fs, ws, cp);
      // alpha'_{t-1} is the next frame's alpha'_{tl+1}: kept in registers across that frame's (light) walk and pass
      // rather than read again
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        au0[j] = al0[j];
        au1[j] = al1[j];
      }
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(q0, bbase, lane16, RES);
      load_chunk(q1, bbase, lane16, RES + 1);
    }
    TC_PSTAMP(1, t, 7)
  };

  // ---- first phase: frames T-1 .. M, pure recursion; rows T .. M+1 stored
  load_chunk(q0, bbase, lane16, RES1);
  load_chunk(q1, bbase, lane16, RES1 + 1);
  request_yn(T - 1);
#ifndef TC_PAIR_NO_PURE
  for (int t = T - 1; t >= M; --t) frame(t, std::integral_constant<int, RES1>(), std::false_type(), std::false_type());
#endif
  {
    // B_M goes to row T + 1; then the hand-off
    const rsrc_t ha = make_rsrc(hist0 + (int64_t)(T + 1) * hist_step, hb0), hb = make_rsrc(hist1 + (int64_t)(T + 1) * hist_step, hb1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        bst4(ha, own16 + j * kPlane, bo0[j]);
        bst4(hb, own16 + j * kPlane, bo1[j]);
      }
  }
  TC_PSTAMP(1, T + 1, 0)
  publish(q.sync + 4 + 2 * pair + 1, tid);
  const bool partner_ok = await(q.sync + 4 + 2 * pair, tid, aRed + kScrAwait);
  TC_PSTAMP(1, T + 1, 1)
  {
    // c_M = 1 / sum_g alpha_M(g) B_M(g), alpha_M = alpha'_M - leaky pi asum_M; then gamma_{M-1} (exp(y_{M-1}) is in LDS,
    // B_M in bo)
    request_au(M - 1);
    request_al(M - 1);
    u4 fs[2];
    f4 ws[2], cp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fs[j] = bld4u(r_fs, own16, j * kPlane);
      ws[j] = bld4(r_ws, own16, j * kPlane);
      cp[j] = leaky * bld4(r_pi, own16, j * kPlane);
    }
    const float asm_a = vload_f32(fn0 + M), asm_b = vload_f32(fn1 + M);
    float d[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < planes) {
        d[0] += hsum((au0[j] - cp[j] * asm_a) * bo0[j]);
        d[1] += hsum((au1[j] - cp[j] * asm_b) * bo1[j]);
      }
    block_sums(d, aRed + kScrDot, wave, lane);
    chat_a = __builtin_amdgcn_rcpf(d[0]);
    chat_b = __builtin_amdgcn_rcpf(d[1]);
    gamma_block(M - 1, fs, ws, cp);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      au0[j] = al0[j];
      au1[j] = al1[j];
    }
  }
  // ---- second phase: frames M-1 .. 0: the derivative row of frame t, gamma_{t-1}
  load_chunk(q0, bbase, lane16, RES2);
  load_chunk(q1, bbase, lane16, RES2 + 1);
#ifndef TC_PAIR_NO_GAMMA
  for (int t = M - 1; t >= 1; --t) frame(t, std::integral_constant<int, RES2>(), std::true_type(), std::false_type());
  frame(0, std::integral_constant<int, RES2>(), std::true_type(), std::true_type());
#endif
  if (!partner_ok && tid == 0) {
    p.seq_ab[s0] = __builtin_nanf("");
    if (valid1) p.seq_ab[s1] = __builtin_nanf("");
  }
}

#ifndef TC_PAIR_RF1
#define TC_PAIR_RF1 0
#endif
#ifndef TC_PAIR_RF2
#define TC_PAIR_RF2 2
#endif
#ifndef TC_PAIR_RB1
#define TC_PAIR_RB1 0
#endif
#ifndef TC_PAIR_RB2
#define TC_PAIR_RB2 0
#endif

template <int PV, bool ACCUM>
__global__ __launch_bounds__(kThreads) void den_tied_pair_kernel(const DenParams p, const PairParams q) {
  // ticket -> (pair, role): whoever starts next becomes the partner of the last unpaired workgroup
  if (threadIdx.x == 0)
    *reinterpret_cast<lds_u *>(q.aRed + kScrTicket) =
        __hip_atomic_fetch_add((gu32 *)q.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t ticket = __builtin_amdgcn_readfirstlane(*reinterpret_cast<lds_u *>(q.aRed + kScrTicket));
  const int pair = (int)(ticket >> 1);
  if (pair >= q.npairs) return;
#ifndef TC_PAIR_ONLY_BWD
  if ((ticket & 1u) == 0u) pair_forward<PV, ACCUM, TC_PAIR_RF1, TC_PAIR_RF2>(p, q, pair);
#endif
#ifndef TC_PAIR_ONLY_FWD
  if ((ticket & 1u) != 0u) pair_backward<PV, ACCUM, TC_PAIR_RB1, TC_PAIR_RB2>(p, q, pair);
#endif
}

struct PairLds {
  uint32_t aGM, aSEC, aRed, aY2, total;
};
PairLds pair_lds(const DenLayout &L, int extra_slots) {
  PairLds o;
  o.aGM = (uint32_t)L.PV * 32u * kThreads + 8u * (uint32_t)L.Hs;
  o.aSEC = o.aGM + 8u * (uint32_t)L.Ps;
  o.aRed = (o.aSEC + 8u * (uint32_t)extra_slots + 15u) & ~15u;
  o.aY2 = o.aRed + kScrBytes;
  o.total = o.aY2 + 8u * kThreads;
  return o;
}

template <int PV>
int launch_pair_v(const DenParams &p, const PairParams &q, int accumulate, size_t lds, hipStream_t stream) {
  void (*k)(const DenParams, const PairParams) = accumulate ? den_tied_pair_kernel<PV, true> : den_tied_pair_kernel<PV, false>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(2 * q.npairs), dim3(kThreads), lds, stream, p, q);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

int pair_norm_stride(int T) { return (T + 2 + 31) & ~31; }
size_t pair_sync_bytes(int S) { return (size_t)(16 + 8 * ((S + 1) / 2) + 255) & ~(size_t)255; }

// The two-sequence form fits: 8 states per thread, at least two frames, and [exp(y) | gather source | gamma |
// secondary rows] of two sequences within one CU's LDS.
bool pair_fits(const DenLayout &L, int extra_slots, int T) {
  if (L.JV != kJvSmall || T < 2) return false;
  return pair_lds(L, extra_slots).total <= (uint32_t)kLdsLimitBytes;
}

int launch_den_tied_pair(const DenParams &p0, int extra_slots, int accumulate, hipStream_t stream) {
  if (!p0.deriv || !p0.pair_norm || !p0.pair_sync || !p0.fwd.cells_pair || !p0.bwd.cells_pair) return TC_ERR_UNSUPPORTED;
  if (!pair_fits(p0.L, extra_slots, p0.T)) return TC_ERR_UNSUPPORTED;
  const PairLds l = pair_lds(p0.L, extra_slots);
  DenParams p = p0;
  PairParams q;
  q.sync = p0.pair_sync;
  q.fwd_cells = p0.fwd.cells_pair;
  q.bwd_cells = p0.bwd.cells_pair;
  q.M = p0.T / 2;
  q.npairs = (p0.S + 1) / 2;
  q.norm_stride = pair_norm_stride(p0.T);
  q.aGM = l.aGM;
  q.aSEC = l.aSEC;
  q.aRed = l.aRed;
  q.aY2 = l.aY2;
  p.fwd_norm = p0.pair_norm;
  p.bwd_norm = p0.pair_norm + (int64_t)(p0.S + 1) * q.norm_stride;  // (S + 1 rows each: the spare row of an odd batch)
  TC_HIP_CHECK(hipMemsetAsync(q.sync, 0, pair_sync_bytes(p0.S), stream));
  const int PV = p.L.PV;
  if (PV == kPvSmall) return launch_pair_v<kPvSmall>(p, q, accumulate, l.total, stream);
  if (PV == kPvMid) return launch_pair_v<kPvMid>(p, q, accumulate, l.total, stream);
  if (PV == kPvLarge) return launch_pair_v<kPvLarge>(p, q, accumulate, l.total, stream);
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
