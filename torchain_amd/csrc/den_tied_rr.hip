// Fused denominator forward-backward for tied graphs WITHOUT hub states, at most 8192 positions and 4096 pdfs -- the
// shape of the headline metric -- with the row sums of the arc walks in REGISTERS and the cell stream ON CHIP.
//
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), reached by the
// reference through src/my_lib_chain.cpp:129-131; the arithmetic, the schedules and the LDS-resident working set are
// those of den_tied_kernel.hip (read that file's header first).  What differs is where the time of that kernel went
// (profiles/r02_phase_stamps.txt, r04_*): 56 % of a frame pair is the two arc walks, and a walk was
//   * issue-bound on its row ends: one scalar test + branch per cell and a commit through LDS per row, and
//   * L1-bound on the 3 of 7 chunks per wave that did not fit the registers: 48 KB per chunk-round through a path
//     that delivers 64 B/clk/CU is 750 cycles, whatever the walk does with the cells.
// Here
//   1. the FMA of a cell accumulates straight into row register number k (GPR-index mode, den_tied_device.h:
//      chunk_rr): a row end is "k += 1" on the scalar unit, the row sums never touch LDS, and the 32 KB of LDS they
//      occupied are free;
//   2. that LDS, plus what the layout had to spare, holds one more chunk of every wave's stream (read back with
//      three ds_read_b128 per lane and frame: 48 KB at 256 B/clk instead of 64), and the registers the row-end
//      bookkeeping and the second stream buffer took hold more resident chunks: RESF = 6 of a wave's 7-8 chunks in
//      the forward phase, RESB = 4 in the backward phase (which carries 24 registers of recursion state);
//   3. a wave walks its stream in the order [chunks from L1 | the LDS chunk | the resident chunks]: the L1 chunks
//      arrive in buffers requested before the frame's barrier and are consumed first, so their registers serve the
//      LDS chunk next.
#include "den_tied_device.h"

namespace tc {

namespace {

// LDS map of this kernel in bytes ([exp(y) | gather source] sit at 0 and kA0 as in den_tied_kernel.hip)
struct RrLayout {
  uint32_t aGM, aAL, aP2, aRed, aAsum, aCells;
  int cells_waves;  // waves 0 .. cells_waves - 1 keep one chunk of their stream in LDS
};

#ifdef TC_RR_AGE_PRIO
#define TC_RR_PRIO(n)
#else
#define TC_RR_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u2 lds_u2;
// LDS chunk: [wa | wb | oc] blocks of 16 waves x 1 KB (the last block only as long as the LDS goes: RrLayout::cells_waves)
constexpr uint32_t kCellsBlock = 1024u * kWaves;
// Row-register images of chunk c of a wave's stream (ScheduleHost::images).  A wave's images live in ONE vector register,
// image word i in lane i (streams of at most 16 chunks), and come out through v_readlane: read from memory where they
// are needed -- s_load_dwordx4 per chunk, or the mask word of den_tied_device.h's walk -- every chunk's block waited for
// a scalar-cache round trip that nothing overlapped (~200 of the ~480 cycles a wave running alone took per chunk).
__device__ __forceinline__ u4 chunk_img(uint32_t vimg, int c) {
#ifdef TC_ABL_NOIMG
  return u4{0xC000C000u, 0xC000C000u, 0xC001C001u, 0xC001C001u};
#endif
  return u4{(uint32_t)__builtin_amdgcn_readlane((int)vimg, 4 * c), (uint32_t)__builtin_amdgcn_readlane((int)vimg, 4 * c + 1),
            (uint32_t)__builtin_amdgcn_readlane((int)vimg, 4 * c + 2), (uint32_t)__builtin_amdgcn_readlane((int)vimg, 4 * c + 3)};
}
constexpr int kRrMaxChunks = 16;

// One walk of a wave's stream: [nl1 chunks through L1 | one chunk from LDS (has_lds) | RES resident chunks] in stream
// order.  Every block carries its own row indices, so the order they are TAKEN in is free, and it is chosen for the L1
// chunks: a chunk-round of all 16 waves is 48 KB through a path of 64 B/clk, so what a wave requests at the start of its
// walk arrives ~1.5 k cycles later, and a wave that needs it sooner sits that out (measured: the first versions of this
// kernel, which took the L1 chunks first, lost everything the faster blocks gained).
//   one L1 buffer (forward):  request #0 | LDS chunk | PRE resident chunks | #0, request #1 | mid() | the other resident
//                             chunks, hook() behind the first | #1 | further L1 chunks one at a time
//   two (backward):           request #0, #1 | LDS chunk | PRE resident chunks | #0, request #2 | mid() | #1, request #3
//                             | the other resident chunks, hook() | #2 | #3 | further L1 chunks
// (the LDS chunk goes through a few transient registers of its own, four cells at a time)
// * the descriptor of the L1 part ends with the wave's last L1 chunk: a request past it costs no traffic, so every load
//   here is UNCONDITIONAL and the compiler can count what is in flight (a wave's vector-memory operations retire in
//   order).  One store behind a condition between a chunk's request and its use turns the counted wait into vmcnt(0),
//   which also waits for that store's acknowledgement (measured: 4.3 k cycles for three chunks); so the frame's deferred
//   stores are unconditional too (hook(): a descriptor of size zero in the frames that have none) and the frame's HBM
//   rows (mid()) are requested behind the cells.
// * issue priority by PROGRESS: a wave that is behind in its walk outranks the waves ahead of it, so the four waves of a
//   SIMD finish together.  (Static priorities by wave age -- den_tied_device.h: age_prio_on -- or none let one wave per
//   SIMD run ahead and leave the last to walk its final chunks alone, every block a serial chain of unpacking, gather
//   latency and FMAs with nothing to overlap it: slowest wave 6.6 k cycles, average 4.8 k.)
template <uint32_t SRC, int RES, int PRE, bool DB, class Mid, class Hook>
__device__ __forceinline__ void walk_rr(const Chunk6 (&res)[RES], rsrc_t l1base, uint32_t lane16, int nl1, bool has_lds, uint32_t cells_addr,
                                        uint32_t mk, RegRows8 &R, Mid mid, Hook hook TC_WALK_ARG) {
  static_assert(PRE < RES, "");
  rows_clear(R);
  nl1 = __builtin_amdgcn_readfirstlane(nl1);
#ifdef TC_ABL_NOL1
  nl1 = 0;
#endif
#ifdef TC_ABL_NOLDSCH
  has_lds = false;
#endif
  const int c0 = nl1 + (has_lds ? 1 : 0);  // the first resident chunk
#ifdef TC_PHASE_STAMPS
  wst[2] = clock64();
#endif
  TC_RR_PRIO(3);
  Chunk6 qa, qb;
  load_chunk(qa, l1base, lane16, 0);
  if constexpr (DB) load_chunk(qb, l1base, lane16, 1);
  if (__builtin_amdgcn_readfirstlane((int)has_lds) != 0) {
    // the LDS chunk, in two halves through a few transient registers: the L1 buffers are in flight.  (The address is
    // formed here, from the lane offset and a scalar: kept in a register of its own across the frame loop it is the first
    // thing the allocator spills, and its reload sits in the walk behind a vmcnt(0).)
    uint32_t ca;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(ca) : "s"(cells_addr), "v"(lane16));
    const u4 img = chunk_img(mk, nl1);
    {
      const u4 w = lds4u(ca);
      const u2 o = *reinterpret_cast<lds_u2 *>(ca + 2 * kCellsBlock);
      quad_rr<SRC>(w, o.x, o.y, img.x, img.y, R);
    }
    {
      const u4 w = lds4u(ca + kCellsBlock);
      const u2 o = *reinterpret_cast<lds_u2 *>(ca + 2 * kCellsBlock + 8);
      quad_rr<SRC>(w, o.x, o.y, img.z, img.w, R);
    }
  }
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    if (i == PRE / 2) TC_RR_PRIO(2);
    chunk_rr<SRC>(res[i], chunk_img(mk, c0 + i), R);
  }
  TC_RR_PRIO(2);
#ifdef TC_PHASE_STAMPS
  {
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
    const long long now = clock64();
    wst[0] += now - wst[2];  // LDS chunk, first resident chunks
    wst[2] = now;
  }
#endif
  if (nl1 > 0) chunk_rr<SRC>(qa, chunk_img(mk, 0), R);
  load_chunk(qa, l1base, lane16, DB ? 2 : 1);
  mid();
  if constexpr (DB) {
    if (nl1 > 1) chunk_rr<SRC>(qb, chunk_img(mk, 1), R);
    load_chunk(qb, l1base, lane16, 3);
  }
#ifdef TC_PHASE_STAMPS
  {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const long long now = clock64();
    wst[1] += now - wst[2];  // L1 chunks
    wst[2] = now;
  }
#endif
  TC_RR_PRIO(1);
#pragma unroll
  for (int i = PRE; i < RES; ++i) {
    chunk_rr<SRC>(res[i], chunk_img(mk, c0 + i), R);
    if (i == PRE) hook();
  }
  TC_RR_PRIO(0);
#ifdef TC_PHASE_STAMPS
  {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const long long now = clock64();
    wst[0] += now - wst[2];  // the other resident chunks
    wst[2] = now;
  }
#endif
  constexpr int kNext = DB ? 2 : 1;
  if (nl1 > kNext) chunk_rr<SRC>(qa, chunk_img(mk, kNext), R);
  if constexpr (DB)
    if (nl1 > 3) chunk_rr<SRC>(qb, chunk_img(mk, 3), R);
  for (int c = 2 * kNext; c < nl1; ++c) {  // (longer streams: one chunk at a time, its load exposed)
    load_chunk(qa, l1base, lane16, c);
    chunk_rr<SRC>(qa, chunk_img(mk, c), R);
  }
#ifdef TC_PHASE_STAMPS
  __builtin_amdgcn_s_waitcnt(0xC07F);
  wst[1] += clock64() - wst[2];  // L1 chunks
#endif
}

template <bool ACCUM, int RESF, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_rr_kernel(const DenParams p, const RrLayout X) {
  constexpr int JV = kJvSmall, PV = kPvSmall;
#ifndef TC_RR_FWD_DB
#define TC_RR_FWD_DB false
#endif
  constexpr bool kFwdDB = TC_RR_FWD_DB;  // two L1 buffers in the forward phase
#ifndef TC_RR_PRE_F
#define TC_RR_PRE_F 3
#endif
#ifndef TC_RR_PRE_B
#define TC_RR_PRE_B 2
#endif
  constexpr int kPreF = TC_RR_PRE_F < RESF ? TC_RR_PRE_F : RESF - 1, kPreB = TC_RR_PRE_B < RESB ? TC_RR_PRE_B : RESB - 1;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int H = p.H, P = p.P, S = p.S, T = p.T;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);  // whole planes of 4096 positions (schedule_owner.cpp: build_owner)
  const uint32_t own16 = 16u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u;                   // exp(y_t)
  constexpr uint32_t kA0 = PV * 16u * kThreads;  // alpha'_t (forward) / Y_t (backward): the gather source
  const uint32_t aGM = X.aGM, aAL = X.aAL, aRed = X.aRed, aAsum = X.aAsum;
  const bool cells_lds = wave < X.cells_waves;
  const uint32_t cells_base = X.aCells + 1024u * (uint32_t)wave;  // (wave-uniform)
  const uint32_t cells_addr = cells_base + lane16;

  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4), row_bytes = 4u * (uint32_t)P;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const float leaky = p.leaky;
  RegRows8 R;
  f4 pi4[JV];
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    pi4[j] = j < planes ? bld4(r_pi, own16, j * kPlane) : mk4(0.f);
    part += hsum(pi4[j]);
  }
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  float asum = block_sum_a(part, aRed, wave, lane);
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist = p.alpha_hist + (int64_t)s * Hs;  // frame t lives at hist + t * hist_step
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
    const int i0 = 4 * (int)tid;
    if (i0 < Ps) {
      const f4 yv = bld4(yrow, own16, 0);
      y2 += hsum(yv * yv);
      lds4_st(kPB + 4u * i0, exp4(yv));
    }
  }
  if (tid == 0) ldsf_st(aAsum, asum);
  float inv_prev = __builtin_amdgcn_rcpf(asum);

  // ---- forward frames t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  {
    const int2 frange = p.fwd.wave_range[wave];
    const int fn = __builtin_amdgcn_readfirstlane(frange.y) / kChunk;
    const rsrc_t fbase = make_rsrc(reinterpret_cast<const char *>(p.fwd.cells) +
                                       (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                                   (uint32_t)(fn + 4) * (3 * 64 * 16));
    const uint32_t fmask = p.fwd.images[(size_t)wave * p.fwd.img_stride * 4 + lane];  // (img_stride >= 16 chunks of 4 words)
    // the stream's last RESF chunks stay in registers for the phase, the one before them in LDS (waves that have a
    // slot there), the rest comes through L1 every frame
    const bool f_lds = cells_lds && fn > RESF;
    const int fnl1 = fn - RESF - (f_lds ? 1 : 0);
    // (the L1 part of the stream under a descriptor of its own: requests past its end cost nothing)
    const rsrc_t fl1 = make_rsrc(reinterpret_cast<const char *>(p.fwd.cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(fnl1 > 0 ? fnl1 : 0) * (3 * 64 * 16));
    Chunk6 fres[RESF];
#pragma unroll
    for (int i = 0; i < RESF; ++i) load_chunk(fres[i], fbase, lane16, fn - RESF + i);
    if (f_lds) {
      Chunk6 q;
      load_chunk(q, fbase, lane16, fnl1);
      *reinterpret_cast<lds_u4 *>(cells_addr) = q.wa;
      *reinterpret_cast<lds_u4 *>(cells_addr + kCellsBlock) = q.wb;
      *reinterpret_cast<lds_u4 *>(cells_addr + 2 * kCellsBlock) = q.oc;
    }
    // The forward phase does not use the gamma / alpha'_{t+1} / second exp(y) regions: when they hold the two
    // per-state tables (C3: exactly), each thread parks its own entries there and the per-state pass reads
    // them at LDS latency instead of waiting for L2 every frame.
    const bool tabs_lds = (X.aRed - X.aGM) >= 8u * (uint32_t)Hs;
    const uint32_t aFS = aGM, aWS = aGM + 4u * (uint32_t)Hs;
    if (tabs_lds) {
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          *reinterpret_cast<lds_u4 *>(aFS + own16 + j * kPlane) = bld4u(r_fs, own16, j * kPlane);
          lds4_st(aWS + own16 + j * kPlane, bld4(r_ws, own16, j * kPlane));
        }
    }
    // Everything requested so far has landed before the first frame.  Without this the compiler guards every use of a
    // resident chunk inside the frame loop with the wait its FIRST iteration needs ("at most N younger operations in
    // flight"), and in steady state that N is smaller than what a frame keeps in flight on purpose -- the HBM row
    // requested under the walk -- so every frame waited for HBM in the middle of its walk.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    TC_STAMP_DECL
    for (int t = 1; t <= T; ++t) {
      TC_STAMP(0)
      __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready
      TC_STAMP(1)
      f4 yreg = mk4(0.f);
#ifdef TC_RR_AGE_PRIO
      age_prio_on(wave);
#endif
      // the history row of frame t-1 is stored from under the walk (den_tied_kernel.hip: a CU issues a 1 KB store
      // instruction only every ~60 cycles): the four wave generations after resident chunk 0, 1, 2, 3
      walk_rr<kA0, RESF, kPreF, kFwdDB>(fres, fl1, lane16, fnl1, f_lds, cells_base, fmask, R, [&]() {
        // y_t (frame T: row T - 1 again, unused) under the rest of the walk
        yreg = bld4(make_rsrc(p.y + ((int64_t)(t < T ? t : T - 1) * S + s) * p.y_stride, row_bytes), own16, 0);
      }, [&]() {
        // the history row of frame t-1 (none at t = 1: a descriptor of size zero drops the stores; the second plane of
        // a graph that has one: past the row's end, dropped too): alpha'_{t-1} of the owned states is still in the gather buffer
        const rsrc_t hist_prev = make_rsrc(hist + (int64_t)(t - 1) * hist_step, t > 1 ? 4u * Hs : 0u);
#pragma unroll
        for (int j = 0; j < JV; ++j) bst4(hist_prev, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
      } TC_WALK_PASS);
      __builtin_amdgcn_s_setprio(0);
      TC_STAMP(2)
      TC_STAMP(3)
      f4 v4[JV];
      part = 0.f;
      u4 fs[JV];
      f4 ws[JV], cpi[JV];
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          fs[j] = tabs_lds ? lds4u(aFS + own16 + j * kPlane) : bld4u(r_fs, own16, j * kPlane);
          ws[j] = tabs_lds ? lds4(aWS + own16 + j * kPlane) : bld4(r_ws, own16, j * kPlane);
          cpi[j] = bld4(r_pi, own16, j * kPlane);  // pi: first touched behind the reduction, which hides its L2 trip
        }
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        v4[j] = mk4(0.f);
        if (j < planes) {
          const f4 F = reg_rows(R, j);
          const f4 al = lds4(kA0 + own16 + j * kPlane);  // alpha'_{t-1} of the owned states
          // alpha_t(g) * asum_{t-1} = p(f(g)) * sum_{h != g} w alpha'_{t-1}(h) + p(s(g)) * w_s * alpha'_{t-1}(g)
          const f4 a = f4{tied_alpha(kPB, fs[j].x, ws[j].x, F.x, al.x), tied_alpha(kPB, fs[j].y, ws[j].y, F.y, al.y),
                          tied_alpha(kPB, fs[j].z, ws[j].z, F.z, al.z), tied_alpha(kPB, fs[j].w, ws[j].w, F.w, al.w)};
          v4[j] = a * inv_prev;
          part += hsum(v4[j]);
        }
      }
      asum = block_sum_a(part, aRed, wave, lane);
      __builtin_amdgcn_sched_barrier(0);  // (keeps the multiply by leaky, and with it the wait for pi, down here)
      TC_STAMP(4)
      float part_tot = 0.f;
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          const f4 a = v4[j] + (leaky * cpi[j]) * asum;
          lds4_st(kA0 + own16 + j * kPlane, a);
          part_tot += hsum(a);
        }
      if (t < T) {
        const int i0 = 4 * (int)tid;
        if (i0 < Ps) {
          y2 += hsum(yreg * yreg);
          lds4_st(kPB + 4u * i0, exp4(yreg));
        }
      }
      if (tid == 0) ldsf_st(aAsum + 4u * t, asum);
      inv_prev = __builtin_amdgcn_rcpf(asum);
      if (t == T) part = part_tot;
    }
    {
      const rsrc_t hist_T = make_rsrc(hist + (int64_t)T * hist_step, 4u * Hs);
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) bst4(hist_T, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
    }
    TC_STAMP(0)
    TC_STAMP_FLUSH(p.stamps)
  }
  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  const float tot = block_sum_a(part, aRed + 4u * kWaves, wave, lane);
  {
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      // [K] log-prob = log(tot) + sum over t < T of log(alpha-sum_t): the scales divided out of frames 1..T
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(ldsf(aAsum + 4u * t));
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d);  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
  }

  // ---- backward   ([K] BetaDashLastFrame, Beta(T), then BetaDashGeneralFrame(t) + Beta(t))
  // beta'_T(h) = 1/tot;  beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The LDS regions now hold
  // Y (gather source), exp(y_t), exp(y_{t-1}), gamma_t and alpha'_{t+1}.
  const float inv_tot = __builtin_amdgcn_rcpf(tot);
  part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j)
    if (j < planes) part += hsum(leaky * bld4(r_pi, own16, j * kPlane)) * inv_tot;
  float bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);  // also orders the reuse of the gather buffer
  f4 areg[JV];
  f4 ycur, ynext;
  f4 bown[JV];  // beta_{t+1} of the owned states (the LDS gather source holds Y instead)
  // two exp(y) buffers: frame t (self-loop terms of the per-state pass) and frame t-1 (written under the arc
  // walk, needed to form Y for the next frame)
  uint32_t pb_cur = kPB, pb_next = X.aP2;
  const int2 brange = p.bwd.wave_range[wave];
  const int bn = __builtin_amdgcn_readfirstlane(brange.y) / kChunk;
  const rsrc_t bbase = make_rsrc(reinterpret_cast<const char *>(p.bwd.cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(brange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(bn + 4) * (3 * 64 * 16));
  const uint32_t bmask = p.bwd.images[(size_t)wave * p.bwd.img_stride * 4 + lane];
  const bool b_lds = cells_lds && bn > RESB;
  const int bnl1 = bn - RESB - (b_lds ? 1 : 0);
  const rsrc_t bl1 = make_rsrc(reinterpret_cast<const char *>(p.bwd.cells) +
                                   (int64_t)(__builtin_amdgcn_readfirstlane(brange.x) / kChunk) * (3 * 64 * 16),
                               (uint32_t)(bnl1 > 0 ? bnl1 : 0) * (3 * 64 * 16));
  Chunk6 bres[RESB];
#pragma unroll
  for (int i = 0; i < RESB; ++i) load_chunk(bres[i], bbase, lane16, bn - RESB + i);
  if (b_lds) {  // (this wave's slots only: no other wave reads them)
    Chunk6 q;
    load_chunk(q, bbase, lane16, bnl1);
    *reinterpret_cast<lds_u4 *>(cells_addr) = q.wa;
    *reinterpret_cast<lds_u4 *>(cells_addr + kCellsBlock) = q.wb;
    *reinterpret_cast<lds_u4 *>(cells_addr + 2 * kCellsBlock) = q.oc;
  }
  {
    const rsrc_t hist_up = make_rsrc(hist + (int64_t)T * hist_step, 4u * Hs);
    const rsrc_t yrow = make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes);
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      bown[j] = mk4(0.f);
      if (j < planes) {
        const int h0 = 4 * ((int)tid + kThreads * j);
        const float b = inv_tot + bsum;
        bown[j] = f4{h0 < H ? b : 0.f, h0 + 1 < H ? b : 0.f, h0 + 2 < H ? b : 0.f, h0 + 3 < H ? b : 0.f};
        lds4_st(aAL + own16 + j * kPlane, bld4(hist_up, own16, j * kPlane));
      }
    }
    const int i0 = 4 * (int)tid;
    ycur = bld4(yrow, own16, 0);
    if (i0 < Ps) {
      lds4_st(pb_cur + 4u * i0, exp4(ycur));
      lds4_st(aGM + 4u * i0, mk4(0.f));
    }
    __syncthreads();  // exp(y_{T-1}) complete: Y_{T-1}(g) = beta_T(g) * p_{T-1}(f(g))
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const u4 fs = bld4u(r_fs, own16, j * kPlane);
        lds4_st(kA0 + own16 + j * kPlane,
                f4{bown[j].x * ldsf(pb_cur + (fs.x & 0xffffu)), bown[j].y * ldsf(pb_cur + (fs.y & 0xffffu)),
                   bown[j].z * ldsf(pb_cur + (fs.z & 0xffffu)), bown[j].w * ldsf(pb_cur + (fs.w & 0xffffu))});
      }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the resident chunks have landed (see the forward phase)
  TC_STAMP_DECL
  for (int t = T - 1; t >= 0; --t) {
    TC_STAMP(0)
    __syncthreads();  // Y, exp(y_t), alpha'_{t+1} ready; gamma zero
    TC_STAMP(1)
    const float asum_t = ldsf(aAsum + 4u * t);
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
    const rsrc_t hist_t = make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs);
#ifdef TC_RR_AGE_PRIO
    age_prio_on(wave);
#endif
    // beta'_t(h) * asum_t = sum over out-arcs of w * Y(dst): the same walk as forward, no atomics.  The derivative row
    // of frame t+1 leaves from under it: it waits, thread-private, in the exp(y) buffer that went dead with frame
    // t+1's per-state pass and that this thread overwrites only after its walk.
    walk_rr<kA0, RESB, kPreB, true>(bres, bl1, lane16, bnl1, b_lds, cells_base, bmask, R, [&]() {
      // frame t-1's y row and alpha'_t of the owned states under the resident part; at t == 0 y re-reads frame 0
      const int tn = t > 0 ? t - 1 : 0;
      ynext = bld4(make_rsrc(p.y + ((int64_t)tn * S + s) * p.y_stride, row_bytes), own16, 0);
#pragma unroll
      for (int j = 0; j < JV; ++j) areg[j] = bld4(hist_t, own16, j * kPlane);  // (a plane the graph does not have: zeros)
    }, [&]() {
      // (no row at t = T - 1: size zero; lanes past the row's end: dropped by the range check)
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t + 1) * S + s) * p.deriv_stride, t < T - 1 ? row_bytes : 0u);
      bst4(drow, own16, lds4(pb_next + own16));
    } TC_WALK_PASS);
    __builtin_amdgcn_s_setprio(0);
    {
      // exp(y_{t-1}) into the other buffer while the slower waves finish their walk
      const int i0 = 4 * (int)tid;
      if (i0 < Ps) lds4_st(pb_next + 4u * i0, exp4(ynext));
    }
    TC_STAMP(2)
    TC_STAMP(3)
    f4 b4[JV];
    uint32_t fpk[JV][2];  // forward-pdf offsets of the owned states, kept for the Y update below
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
    const float asum_up = ldsf(aAsum + 4u * (t + 1));
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      b4[j] = mk4(0.f);
      if (j < planes) {
        const u4 fs = bld4u(r_fs, own16, j * kPlane);
        const f4 ws = bld4(r_ws, own16, j * kPlane);
        const f4 cp = leaky * bld4(r_pi, own16, j * kPlane);
        f4 a = reg_rows(R, j);
        const f4 al = areg[j];  // alpha'_t of the owned states
        const f4 aup = lds4(aAL + own16 + j * kPlane);
        // Everything the arcs INTO an owned state g contribute to gamma_t, from per-state quantities (den_tied_kernel.hip):
        //   self-loop arc : occ_s = w_s * beta_{t+1}(g) * p_t(s(g)) * alpha'_t(g) / asum_t   -> gamma_t(s(g))
        //   forward class : occ_f = beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart)               -> gamma_t(f(g))
        // with alpha_{t+1} = alpha'_{t+1} - leaky*pi*asum_{t+1}; the self-loop arc also adds
        // vf_s = w_s * beta_{t+1}(g) * p_t(s(g)) to beta'_t(g) * asum_t.
        auto one = [&](uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx, float ax) {
          const float ps_ws = ldsf(pb_cur + (fsx >> 16)) * wsx;
          const float selfpart = ps_ws * alx * inv_as;  // self-loop part of alpha_{t+1}(g)
          const float bos = kGammaScale * bo;            // power-of-two scale: exact
          gamma_add_a(aGM + (fsx >> 16), bos * selfpart);
          gamma_add_a(aGM + (fsx & 0xffffu), bos * fmaxf((aupx - cpx * asum_up) - selfpart, 0.f));
          return fmaf(ps_ws, bo, ax);                    // vf_s into beta'_t(g) * asum_t
        };
        a.x = one(fs.x, ws.x, bown[j].x, al.x, aup.x, cp.x, a.x);
        a.y = one(fs.y, ws.y, bown[j].y, al.y, aup.y, cp.y, a.y);
        a.z = one(fs.z, ws.z, bown[j].z, al.z, aup.z, cp.z, a.z);
        a.w = one(fs.w, ws.w, bown[j].w, al.w, aup.w, cp.w, a.w);
        b4[j] = a * inv_as;  // [K] * inv_arbitrary_scale
        fpk[j][0] = (fs.x & 0xffffu) | (fs.y << 16);
        fpk[j][1] = (fs.z & 0xffffu) | (fs.w << 16);
        part += hsum(cp * b4[j]);
        if (t == 0) part_ab += hsum(al * b4[j]);
      }
    }
    bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    TC_STAMP(4)
    {
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, row_bytes);
      const int i0 = 4 * (int)tid;
      if (i0 < Ps) {
        const u4 gu = lds4u(aGM + 4u * i0);
        lds4_st(aGM + 4u * i0, mk4(0.f));
        const f4 g = f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} * kGammaInvScale;
        if (t == 0) part_g += hsum(g);
        f4 o = p.deriv_weight * g - p.l2_scale * ycur;
        if (ACCUM) o += bld4(drow, own16, 0);
        if (t > 0)
          lds4_st(pb_cur + 4u * i0, o);
        else
          bst4(drow, own16, o);
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum_a(part_ab, aRed + 4u * kWaves, wave, lane);
      const float gs = block_sum_a(part_g, aRed + 8u * kWaves, wave, lane);
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
    // beta_t = beta'_t + leaky-sum; next frame's gather source Y_{t-1} = beta_t * p_{t-1}(f)
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const f4 b = b4[j] + bsum;
        bown[j] = b;
        const f4 yv = f4{b.x * ldsf(pb_next + (fpk[j][0] & 0xffffu)), b.y * ldsf(pb_next + (fpk[j][0] >> 16)),
                         b.z * ldsf(pb_next + (fpk[j][1] & 0xffffu)), b.w * ldsf(pb_next + (fpk[j][1] >> 16))};
        lds4_st_at(own16, kA0 + j * kPlane, yv);
        lds4_st(aAL + own16 + j * kPlane, areg[j]);
      }
    ycur = ynext;
    const uint32_t tmp = pb_cur;
    pb_cur = pb_next;
    pb_next = tmp;
  }
}

#ifndef TC_RR_RESF
#define TC_RR_RESF 5
#endif
#ifndef TC_RR_RESB
#define TC_RR_RESB 3
#endif
static_assert(TC_RR_RESF <= kTiedMinChunks && TC_RR_RESB <= kTiedMinChunks, "the resident chunks must exist in every wave's stream");

bool rr_layout(const DenLayout &L, int T, RrLayout *X) {
  if (L.JV != kJvSmall || L.PV != kPvSmall || !L.alpha_in_lds) return false;
  const uint32_t p_floats = (uint32_t)L.PV * 4u * kThreads;
  uint32_t off = p_floats + (uint32_t)L.Hs;  // [exp(y) | gather source]
  X->aGM = 4u * off;
  off += (uint32_t)L.Ps;
  X->aAL = 4u * off;
  off += (uint32_t)L.Hs + 4u;
  X->aP2 = 4u * off;
  off += p_floats;
  X->aRed = 4u * off;
  off += 4u * kWaves;
  X->aAsum = 4u * off;
  off += (uint32_t)round4(T + 1);
  X->aCells = 4u * off;
  if (X->aCells > (uint32_t)kLdsLimitBytes) return false;
  const uint32_t room = (uint32_t)kLdsLimitBytes - X->aCells;
  X->cells_waves = room > 2u * kCellsBlock ? std::min<int>(kWaves, (int)((room - 2u * kCellsBlock) / 1024u)) : 0;
  return true;
}

}  // namespace

// Graphs and batches this kernel takes: tied, laid out for 8 states and 4 pdfs per thread with the roomy LDS
// layout, no hub states in either direction (their secondary rows commit to LDS slots: den_tied_kernel.hip), rows of y
// and of the derivative 16-byte aligned (one load / store instruction per row on every path: counted waits).
bool rr_fits(const DenParams &p) {
  RrLayout X;
  return p.deriv != nullptr && p.fwd.images && p.bwd.images && p.fwd.max_chunks <= kRrMaxChunks && p.bwd.max_chunks <= kRrMaxChunks && p.y_vec && p.d_vec && p.fwd.nfix == 0 && p.bwd.nfix == 0 && debug_flag(kDbgRegRows) && rr_layout(p.L, p.T, &X);
}

int launch_den_tied_rr(const DenParams &p, int accumulate, hipStream_t stream) {
  RrLayout X;
  if (!rr_layout(p.L, p.T, &X)) return TC_ERR_UNSUPPORTED;
  const size_t lds = (size_t)X.aCells + (X.cells_waves ? 2u * kCellsBlock + (size_t)X.cells_waves * 1024u : 0u);
  void (*k)(const DenParams, const RrLayout) =
      accumulate ? den_tied_rr_kernel<true, TC_RR_RESF, TC_RR_RESB> : den_tied_rr_kernel<false, TC_RR_RESF, TC_RR_RESB>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds, stream, p, X);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
