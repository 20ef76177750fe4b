// Fused denominator forward-backward for TIED graphs of 16385..40960 positions on gfx950: the "plane-wise" form of
// den_tied_kernel.hip (same mapping: one workgroup = one sequence = one CU, owner-computes schedules, the per-state
// formulas of den_tied_frames.h).  This is the size class of the den.fst the reference's recipe really loads (a pruned
// phone LM with Kaldi's default 2000 extra LM states: example/chime5/train_faster.py:91 -> src/my_lib_example.cpp:129-134
// takes any FST), which until round 5 fell onto the streamed kernels (den_slab_kernel.hip: 18 ms per 256 x 150 batch
// against 1.4 ms for the 8192-state C3 graph).
//
// What changes beyond 16384 positions.  The gather source alone is 4 bytes per position of the CU's 160 KB (96 KB at
// 24576 positions): neither the row sums of all states (another 4 bytes per position) nor a second exp(y) buffer nor
// parked alpha' fit, and a thread owns 20-28 states -- held in registers the way the smaller instantiations hold them
// (alpha_t, beta_{t+1}, alpha'_t, beta'_t, the forward pdfs) they would be ~100 registers, and the walk would have none.
// So:
//   * the frame is taken a PLANE (one float4 of states per thread, 4096 positions) at a time.  The schedule
//     (schedule_owner.cpp, `planewise`) cuts every wave's stream at chunk boundaries into its secondary rows and one
//     sub-stream per plane; when a plane's sub-stream ends, the wave runs that plane's per-state pass and walks on.  The
//     row sums of all planes share four accumulator rows per wave.
//   * per-state values cross the walks in as FEW registers as a frame can do with: one float4 per plane (24-28 registers) --
//     alpha_t as the passes form it, until the frame's tail adds the leaky term; beta_{t+1} / beta'_t / beta_t through a
//     backward frame -- indexed by the wave-uniform run-time plane number through opaque selects (rget / rset).  Everything
//     else a plane's pass needs (tables, alpha_t and alpha_{t+1} of the backward pass) is requested from L2 a sub-stream
//     ahead.  Round 5 kept NOTHING per-state in registers: alpha_t was read back from the history row it had just been
//     written to, and beta'_t lived in one more row of the history buffer -- a written and two read rows per frame and CU
//     whose 32 copies per XCD took 3 of its L2's 4 MB from the cell stream (R4: 9.5 ms per 256 x 150 batch; now 7.55).
//   * the rest of the registers go to the stream: two chunk buffers in the fused kernel, three in the two-workgroup form.
//     The history rows are stored `nt`: their next reader is the backward pass.
//   * cells carry 16-bit POSITIONS (byte offset = one SDWA shift), one row-end byte per chunk.
// One instantiation serves 5 and 6 planes, one 7 (the plane index is a wave-uniform run-time value), one 8 to 10: beyond 28672
// positions the gather source is in LDS a half at a time and a frame is two walks (SPLIT, below).
//
// Batches of at most half the CUs run TWO workgroups per sequence that meet in the middle (den_tied_mitm.hip's scheme
// with these frames): role F = the forward frame, from the meeting frame on forming gamma_{t-1} from the stored B_t;
// role B = the backward frame with normalisers of its own and no gamma down to the meeting frame, then as in the fused
// kernel.  forward_frame<GAMMA> / backward_frame<PURE> below.
#include "den_handover.h"
#include "den_tied_frames.h"

namespace tc {

namespace {

// SPLIT: graphs of 28673..40960 positions -- the gather source is in LDS a HALF at a time (the first ceil(planes / 2) planes of
// positions, then the others); a frame is two walks: the first over the cells whose source lies in the first half, whose
// per-plane "pass" only parks the plane's row sums in the workspace, then -- the second half of the source brought in from
// the workspace, where the previous frame's tail left it -- the second, whose per-plane pass adds the parked sums and is
// the frame's real per-state pass.  Two more barriers and ~0.5 MB of L2 traffic per frame pair and CU beside 4.8 MB of cells.
// REGS (split source, fused kernel): what waits between a frame's two walks waits in the per-state registers where it can --
// the forward walk's row sums in R (free until the second walk's passes fill it), the second half of the backward gather
// source re-formed from R's beta rows -- instead of in the workspace: X2 at 256 sequences 18.5 -> 17.5 ms.  The two-workgroup
// form keeps the workspace rows: with them in registers its kernel (both roles' code) spills 80 registers and loses 25 %.
template <bool ACCUM, int MAXP, int BUF, bool SPLIT = false, bool REGS = false>
struct PlaneSeq {
  static constexpr uint32_t kPB = 0u;             // exp(y_t)
  static constexpr uint32_t kA0 = 16u * kThreads;  // alpha'_t (forward) / Y_t (backward): the gather source
  static constexpr int kMaxPlanes = MAXP;          // instantiated for graphs of up to 6 and of 7 planes

  const DenParams &p;
  const uint32_t tid, lane;
  const int wave, s;
  const int H, S, T, Hs, Ps, planes;
  const uint32_t own16, lane16;
  const uint32_t aACC, vrow, aGM, aRed, aAsum;
  const AsumRow asums;
  const uint32_t row_bytes;
  const rsrc_t r_pi, r_fs, r_ws;
  const float leaky;
  const int64_t hist_step;
  float *const hist;     // alpha history (un-dashed): frame t at hist + t * hist_step
  // two-workgroup form only (null in the fused kernel)
  float *const fn;       // [T + 2] asum_0..T (role F writes, role B reads)
  float *const bn;       // [T + 1] role B's normalisers
  float *const bhist;    // B history
  // ---- the running direction's stream (wave-uniform first)
  int total, nfix;
  uint32_t sec;
  rsrc_t base;
  const int32_t *fix_begin;
  const int2 *fix;
  // ---- per lane
  uint32_t vmask, ends;
  float asum, inv_prev, bsum, part, part_tot, y2;
  float chat;  // c^_t: the scale the fixed-point adds of the running GAMMA frame use
  f4 bt_n;     // GAMMA frames: B_t of the next plane's owned states
  // The frame's own per-state values of ALL planes, in registers across the walks (round 6): alpha_t as the passes form it
  // (forward: the frame's tail adds the leaky term without reading its history row back), beta_{t+1} on entry to a backward
  // frame / beta'_t behind a plane's pass / beta_t behind the tail.  The plane index is a wave-uniform run-time value: a
  // plane's float4 goes in and out through selects (rget / rset: ~50 v_cndmask per pass) -- no indexed register file
  // access, nothing for the compiler to put in scratch.  Round 5 kept beta'_t in one more row of the history buffer:
  // a written and two read rows per frame and CU, and 32 such rows per XCD took 3 of its L2's 4 MB from the cell stream.
  f4 R[MAXP];
  // the next plane's tables and history values, requested a plane ahead
  u4 fs_n;
  f4 ws_n, cp_n, al_n, aup_n;
  int fx0_n, fx1_n;
  // split gather source only (an empty struct otherwise: members the other instantiations must not carry -- new members next
  // to the old ones changed how the compiler packs them and cost every instantiation 5 registers)
  struct SplitOn {
    int planes_a;    // planes of the gather source in LDS at a time: the first half's
    float *src_b;    // the second half of the running frame's gather source (workspace)
    float *parked;   // the first walk's row sums, [position] (workspace)
    f4 part_n;       // the next plane's parked row sums
    f4 al_n;         // forward: the next plane's parked alpha'_{t-1} (planes of the first half)
  };
  struct SplitOff {};
  std::conditional_t<SPLIT, SplitOn, SplitOff> sp;
#ifdef TC_PHASE_STAMPS
  long long st_prev, st_acc[8];
#endif

  // role: which of a sequence's two workgroups (two-workgroup form; the split source's scratch rows are per workgroup)
  __device__ __forceinline__ PlaneSeq(const DenParams &pp, int seq, int role = 0)
      : p(pp), tid(threadIdx.x), lane(threadIdx.x & 63u), wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)), s(seq), H(pp.H), S(pp.S),
        T(pp.T), Hs(pp.L.Hs), Ps(pp.L.Ps), planes(pp.L.Hs / (4 * kThreads)), own16(16u * threadIdx.x), lane16(16u * (threadIdx.x & 63u)),
        aACC(4u * (uint32_t)pp.L.off_acc),
        vrow(4u * (uint32_t)pp.L.off_acc + 1024u * (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 4u * (threadIdx.x & 63u)),
        aGM(4u * (uint32_t)pp.L.off_g), aRed(4u * (uint32_t)pp.L.off_red), aAsum(4u * (uint32_t)pp.L.off_asum), asums(pp, seq), row_bytes(4u * (uint32_t)pp.P),
        r_pi(make_rsrc(pp.pi, 4u * (uint32_t)(pp.L.Hs + 4))), r_fs(make_rsrc(pp.tied_fs, 4u * (uint32_t)(pp.L.Hs + 4))),
        r_ws(make_rsrc(pp.tied_w, 4u * (uint32_t)(pp.L.Hs + 4))), leaky(pp.leaky), hist_step((int64_t)pp.S * pp.L.Hs),
        hist(pp.alpha_hist + (int64_t)seq * pp.L.Hs), fn(pp.fwd_norm ? pp.fwd_norm + (int64_t)seq * (pp.T + 2) : nullptr),
        bn(pp.bwd_norm ? pp.bwd_norm + (int64_t)seq * (pp.T + 1) : nullptr),
        bhist(pp.beta_hist ? pp.beta_hist + (int64_t)seq * pp.L.Hs : nullptr) {
    if constexpr (SPLIT) {
      sp.planes_a = pp.L.src_planes;
      sp.src_b = pp.src_scratch + (int64_t)(2 * seq + role) * (4 * kThreads) * (pp.L.JV - pp.L.src_planes);
      sp.parked = pp.part_scratch + (int64_t)(2 * seq + role) * (pp.L.Hs + 4 * kThreads * pp.L.src_planes);
    }
  }

  __device__ __forceinline__ rsrc_t hist_row(int t) const { return make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs); }
  __device__ __forceinline__ rsrc_t bhist_row(int t) const { return make_rsrc(bhist + (int64_t)t * hist_step, 4u * Hs); }
  __device__ __forceinline__ bool own_pdfs() const { return 4 * (int)tid < Ps; }
  // (opaque to the optimiser: written as C selects, the chains become a switch on the plane index whose cases each carry a
  // copy of the pass, and the walk's scalar loop control ends up in vector registers)
  // dst = j == k ? a : dst, j wave-uniform
  static __device__ __forceinline__ void cmov4(int j, int k, f4 &dst, f4 a) {
    float x = dst.x, y = dst.y, z = dst.z, w = dst.w;
    asm volatile("s_cmp_eq_u32 %8, %9\n\ts_cselect_b64 vcc, -1, 0\n\t"
                 "v_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %5, vcc\n\t"
                 "v_cndmask_b32_e32 %2, %2, %6, vcc\n\tv_cndmask_b32_e32 %3, %3, %7, vcc"
                 : "+v"(x), "+v"(y), "+v"(z), "+v"(w)
                 : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "s"(j), "n"(k)
                 : "vcc", "scc");
    dst = f4{x, y, z, w};
  }
  __device__ __forceinline__ f4 rget(int j) const {
    const int ju = __builtin_amdgcn_readfirstlane(j);
    f4 v = R[0];
#pragma unroll
    for (int k = 1; k < MAXP; ++k) cmov4(ju, k, v, R[k]);
    return v;
  }
  __device__ __forceinline__ void rset(int j, f4 v) {
    const int ju = __builtin_amdgcn_readfirstlane(j);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) cmov4(ju, k, R[k], v);
  }

  __device__ __forceinline__ void stamps_reset() {
#ifdef TC_PHASE_STAMPS
    st_prev = clock64();
    for (int i = 0; i < 8; ++i) st_acc[i] = 0;
#endif
  }
  __device__ __forceinline__ void stamps_flush(int at) {
#ifdef TC_PHASE_STAMPS
    TC_STAMP(0)
    if (blockIdx.x == 0 && lane == 0)
      for (int i = 0; i < 8; ++i) p.stamps[at + wave * 8 + i] = st_acc[i];
#endif
  }

  __device__ __forceinline__ void xent_zero_row(int t) {
    if (p.xent_zero && own_pdfs())
      row_st(make_rsrc(p.xent_zero + ((int64_t)t * S + s) * p.xent_stride, row_bytes), own16, p.x_vec, mk4(0.f));
  }

  // ---- one direction's stream: descriptor, row-end bytes (four chunks to a word, word i in lane i), the chunk at which each
  // sub-stream ends (sub-stream i in lane i), the wave's first secondary row
  __device__ __forceinline__ void stream_begin(const ScheduleDev &sc) {
    const int subs = (SPLIT ? 2 : 1) * (planes + 1);
    const int2 r0 = sc.wave_range[wave * subs], r1 = sc.wave_range[wave * subs + subs - 1];
    total = __builtin_amdgcn_readfirstlane(r1.y) / kChunk;
    // (the descriptor covers the look-ahead past the wave's last chunk: the array ends with readable padding)
    base = make_rsrc(reinterpret_cast<const char *>(sc.cells) + (int64_t)(__builtin_amdgcn_readfirstlane(r0.x) / kChunk) * (3 * 64 * 16),
                     (uint32_t)(total + kBuffers) * (3 * 64 * 16));
    vmask = sc.masks[(size_t)wave * sc.mask_stride + lane];  // (the array ends with a register's worth of padding)
    ends = (int)lane < subs ? (uint32_t)(sc.wave_range[wave * subs + (int)lane].y / kChunk) : 0u;
    sec = aACC + 256u * (uint32_t)(4 * kWaves + sc.extra_first[wave]);
    nfix = sc.nfix;
    fix_begin = sc.fix_begin;
    fix = sc.fix;
  }
  __device__ __forceinline__ uint32_t mask_of(int c) const {
    return (uint32_t)__builtin_amdgcn_readlane((int)vmask, c >> 2) >> (8 * (c & 3));
  }

  // The frame's walk: chunks through kBuffers buffers, kBuffers - 1 requested ahead; whenever the sub-stream of a plane ends (at chunk
  // boundaries, by construction) `pass(plane)` runs.  q0 / q1 arrive requested (chunks 0 and 1: ahead of the frame's barrier).
#ifndef TC_PW_BUFFERS_FUSED
#define TC_PW_BUFFERS_FUSED 2
#endif
#ifndef TC_PW_BUFFERS_PAIR
#define TC_PW_BUFFERS_PAIR 3
#endif
#ifndef TC_PW_HIST_AUX
#define TC_PW_HIST_AUX 2  /* cache policy of the alpha history's stores (0: default, 2: nt: the backward pass is their next reader -- R4 8.04 -> 7.80 ms) */
#endif
#ifndef TC_PW_BWD_PRIO
#define TC_PW_BWD_PRIO 0  /* 1: the backward walks run at issue priorities by wave age, youngest highest; 2: the forward walks too */
#endif
#ifndef TC_PW_AL_AUX
#define TC_PW_AL_AUX 0  /* ... of the backward pass's first read of a history row (alpha_t in frame t) */
#endif
#ifndef TC_PW_AUP_AUX
#define TC_PW_AUP_AUX 0  /* ... of its second and last read (alpha_{t+1} in frame t) */
#endif
  // chunk buffers: kBuffers - 1 chunks requested ahead.  Two in the fused kernel, three in the two-workgroup form (measured in
  // round 6, with 24-28 registers of per-state rows across the walks: R4 at 256 sequences 7.76 -> 7.55 ms with two, but at 128
  // -- two workgroups per sequence, half the CUs' L1s on one stream each -- 4.24 -> 4.89)
  static constexpr int kBuffers = BUF;
  // (the first requests of a frame, ahead of its barrier)
  // (half: which of a split source's two walks; its first chunk is where the first walk's last sub-stream ends)
  __device__ __forceinline__ int first_chunk(int half) const {
    return SPLIT && half ? __builtin_amdgcn_readlane((int)ends, planes) : 0;
  }
  __device__ __forceinline__ void request_first(Chunk6 (&q)[kBuffers], int half = 0) {
    const int c0 = first_chunk(half);
#pragma unroll
    for (int k = 0; k + 1 < kBuffers; ++k) load_chunk(q[k], base, lane16, c0 + k);
  }
  template <class Pass>
  __device__ __forceinline__ void run_stream(Chunk6 (&q)[kBuffers], Pass pass, int half = 0) {
    if constexpr (!SPLIT) {
      // (kept apart from the general form below, word for word as it was: stated through first_chunk() / c_end the same loop
      // costs the one-walk instantiations five registers)
      int c = 0, sub = 0;
      int next_end = __builtin_amdgcn_readlane((int)ends, 0);
      float acc = 0.f;
      const RowCommit plane_rows{aACC + 1024u * (uint32_t)wave, sec, 4};
      RowCommit rc{sec, sec, 1 << 30};  // sub-stream 0: the wave's secondary rows, private slots one after the other
      if (next_end == 0) {              // (none)
        sub = 1;
        rc = plane_rows;
        next_end = __builtin_amdgcn_readlane((int)ends, 1);
      }
      while (c < total) {
#pragma unroll
        for (int k = 0; k < kBuffers; ++k) {
          load_chunk(q[(k + kBuffers - 1) % kBuffers], base, lane16, c + kBuffers - 1);
          chunk_pw<kA0>(q[k], mask_of(c), acc, rc);
          ++c;
          if (c == next_end) {  // the sub-stream of a plane (or the secondary rows) ends here
            if (sub > 0) pass(sub - 1);
            ++sub;
            rc = plane_rows;
            next_end = sub > planes ? -1 : __builtin_amdgcn_readlane((int)ends, sub <= planes ? sub : 0);
          }
          if (c >= total) break;
        }
      }
    } else {
      const int s0 = half ? planes + 1 : 0;  // the walk's first sub-stream
      int c = first_chunk(half), sub = 0;
      const int c_end = __builtin_amdgcn_readlane((int)ends, s0 + planes);
      int next_end = __builtin_amdgcn_readlane((int)ends, s0);
      float acc = 0.f;
      const RowCommit plane_rows{aACC + 1024u * (uint32_t)wave, sec, 4};
      RowCommit rc{sec, sec, 1 << 30};
      if (next_end == c) {  // (no secondary rows)
        sub = 1;
        rc = plane_rows;
        next_end = __builtin_amdgcn_readlane((int)ends, s0 + 1);
      }
      while (c < c_end) {
#pragma unroll
        for (int k = 0; k < kBuffers; ++k) {
          load_chunk(q[(k + kBuffers - 1) % kBuffers], base, lane16, c + kBuffers - 1);
          chunk_pw<kA0>(q[k], mask_of(c), acc, rc);
          ++c;
          if (c == next_end) {
            if (sub > 0) pass(sub - 1);
            ++sub;
            rc = plane_rows;
            next_end = sub > planes ? -1 : __builtin_amdgcn_readlane((int)ends, s0 + (sub <= planes ? sub : 0));
          }
          if (c >= c_end) break;
        }
      }
    }
  }

  // the fix-up list of (this thread, plane[, half]): secondary rows of its hub states, folded into the plane's row sums
  __device__ __forceinline__ void request_fix(int j, int half = 0) {
    fx0_n = fx1_n = 0;
    if (nfix) {
      const int at = ((SPLIT && half ? kThreads : 0) + (int)tid) * planes + j;
      fx0_n = fix_begin[at];
      fx1_n = fix_begin[at + 1];
    }
  }

  // ---- split gather source: the first walk of a frame (both directions).  Its per-plane pass parks the plane's row sums;
  // behind it the second half of the source replaces the first in LDS (two barriers: every wave has left the first half,
  // the second is complete).  q leaves requested for the second walk.
  // FORWARD: the forward per-state pass also needs alpha'_{t-1} of the OWNED states (the self-loop term); for the planes of
  // the first half that is in LDS only now, so it is parked too.
  template <bool FORWARD>
  __device__ __forceinline__ void first_walk(Chunk6 (&q)[kBuffers]) {
    const rsrc_t r_part = part_rsrc();
    run_stream(q, [&](int j) __attribute__((always_inline)) {
      const int fx0 = fx0_n, fx1 = fx1_n;
      for (int e = fx0; e < fx1; ++e) fold_row_pw(fix[e], vrow, aACC, Hs);
      // (forward: the per-state rows R are free until the second walk's passes fill them with alpha_t -- the row sums wait there
      // instead of going through L2: 320 KB per frame and CU whose 32 copies per XCD do not fit its L2 anyway)
      if (FORWARD && REGS)
        rset(j, own_rows(vrow, 0));
      else
        bst4_aux<0>(r_part, own16 + (uint32_t)j * kPlane, own_rows(vrow, 0));
      if (FORWARD && j < sp.planes_a) bst4_aux<0>(r_part, own16 + (uint32_t)j * kPlane + 4u * (uint32_t)Hs, lds4(kA0 + own16 + (uint32_t)j * kPlane));
      request_fix(j + 1 < planes ? j + 1 : j, 0);
    }, 0);
    request_first(q, 1);
    if (FORWARD || !REGS) {
      const rsrc_t r_src = make_rsrc(sp.src_b, 16u * kThreads * (uint32_t)(planes - sp.planes_a));
      f4 sb[(MAXP + 1) / 2];
#pragma unroll
      for (int j = 0; j < (MAXP + 1) / 2; ++j) sb[j] = bld4(r_src, own16, j * kPlane);  // (beyond the half's planes: zeros)
      __syncthreads();  // every wave has gathered its last value of the first half
#pragma unroll
      for (int j = 0; j < (MAXP + 1) / 2; ++j)
        if (j < planes - sp.planes_a) lds4_st(kA0 + own16 + j * kPlane, sb[j]);
    } else {
      // backward: the second half of Y_t = beta_{t+1} * p_t(f) is formed here, from the beta rows the registers hold and the
      // exp(y_t) buffer, instead of travelling through the workspace (a written and a read row of per-CU data per frame)
      u4 fsb[(MAXP + 1) / 2];
#pragma unroll
      for (int j = 0; j < (MAXP + 1) / 2; ++j) fsb[j] = bld4u(r_fs, own16, (uint32_t)(sp.planes_a + j) * kPlane);
      __syncthreads();  // every wave has gathered its last value of the first half
#pragma unroll
      for (int j = 0; j < (MAXP + 1) / 2; ++j)
        if (j < planes - sp.planes_a) {
          const f4 b = rget(sp.planes_a + j);
          lds4_st(kA0 + own16 + j * kPlane, f4{b.x * ldsf(kPB + (fsb[j].x & 0xffffu)), b.y * ldsf(kPB + (fsb[j].y & 0xffffu)),
                                               b.z * ldsf(kPB + (fsb[j].z & 0xffffu)), b.w * ldsf(kPB + (fsb[j].w & 0xffffu))});
        }
    }
    __syncthreads();  // the second half is in place
  }
  // a plane's value of the next frame's gather source: the first half's planes to LDS, the others to the workspace
  __device__ __forceinline__ void source_st(int j, f4 v) {
    if constexpr (SPLIT) {
      if (j < sp.planes_a)
        lds4_st(kA0 + own16 + j * kPlane, v);
      else
        bst4_aux<0>(make_rsrc(sp.src_b, 16u * kThreads * (uint32_t)(planes - sp.planes_a)), own16 + (uint32_t)(j - sp.planes_a) * kPlane, v);
    } else {
      lds4_st(kA0 + own16 + j * kPlane, v);
    }
  }
  // backward: only the first half's planes are stored (the second half of Y is formed from the beta rows when it is needed)
  __device__ __forceinline__ void source_st_bwd(int j, f4 v) {
    if constexpr (SPLIT && REGS) {
      if (j < sp.planes_a) lds4_st(kA0 + own16 + j * kPlane, v);
    } else {
      source_st(j, v);
    }
  }
  // ... scaled in place (the two-workgroup form's hand-over)
  __device__ __forceinline__ void source_scale(int j, float c) {
    if constexpr (SPLIT) {
      if (j >= sp.planes_a) {
        if constexpr (!REGS) {
          const rsrc_t r = make_rsrc(sp.src_b, 16u * kThreads * (uint32_t)(planes - sp.planes_a));
          const uint32_t at = own16 + (uint32_t)(j - sp.planes_a) * kPlane;
          bst4_aux<0>(r, at, bld4(r, at, 0) * c);
        }
        return;  // (REGS: the second half of Y is formed from the beta rows, which the caller scales)
      }
    }
    lds4_st(kA0 + own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane) * c);
  }
  __device__ __forceinline__ rsrc_t part_rsrc() const {
    if constexpr (SPLIT)  // [row sums of every plane | alpha'_{t-1} of the first half's planes]
      return make_rsrc(sp.parked, 4u * (uint32_t)Hs + 16u * kThreads * (uint32_t)sp.planes_a);
    else
      return make_rsrc(hist, 0u);
  }

  // ================================================================================================== forward
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  __device__ __forceinline__ void forward_begin() {
    part = 0.f;
    for (int j = 0; j < planes; ++j) part += hsum(bld4(r_pi, own16, j * kPlane));
    asum = block_sum_a(part, aRed, wave, lane);
    const rsrc_t h0 = hist_row(0);
    for (int j = 0; j < planes; ++j) {
      const f4 pi4 = bld4(r_pi, own16, j * kPlane);
      bst4_aux<0>(h0, own16 + j * kPlane, pi4);  // the history keeps alpha UN-dashed
      source_st(j, pi4 + (leaky * pi4) * asum);
    }
    y2 = 0.f;
    if (own_pdfs()) {
      const f4 yv = row_ld(make_rsrc(p.y + (int64_t)s * p.y_stride, row_bytes), own16, p.y_vec);
      y2 = hsum(yv * yv);
      lds4_st(kPB + own16, exp4(yv));
    }
    if (tid == 0) {
      asums.st(0, asum);
      if (fn) fn[0] = asum;
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) R[k] = mk4(0.f);
    inv_prev = __builtin_amdgcn_rcpf(asum);
    part_tot = 0.f;
    chat = 0.f;
    bt_n = mk4(0.f);
    stream_begin(p.fwd);
    stamps_reset();
  }

  // frame t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t)).  GAMMA (role F behind the meeting frame): the two parts of
  // alpha_t(g) times the stored B_t(g) are the occupations in frame t-1 of the forward-class arcs into g and of its self-loop
  template <bool GAMMA>
  __device__ __forceinline__ void forward_frame(int t) {
    Chunk6 q[kBuffers];
    request_first(q);
    if (!SPLIT) {
      fs_n = bld4u(r_fs, own16, 0);
      ws_n = bld4(r_ws, own16, 0);
    }
    request_fix(0);
    const rsrc_t brow = GAMMA ? bhist_row(t) : make_rsrc(hist, 0u);
    if (GAMMA) bt_n = bld4(brow, own16, 0);
    float n_t = 1.f;
    if (GAMMA && t < T) n_t = vload_f32(bn + t);  // for c^_{t+1}
    const float gs = kGammaScale * chat;
    float dpart = 0.f;
    __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready
    TC_STAMP(0)
    f4 yreg = mk4(0.f);
    if (t < T) yreg = row_ld(make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes), own16, p.y_vec);  // y_t under the walks
    const rsrc_t hist_t = hist_row(t);
    const rsrc_t r_part = part_rsrc();
    if constexpr (SPLIT) {
      first_walk<true>(q);  // the cells whose source lies in the first half; the second half of alpha'_{t-1} is in LDS behind it
      fs_n = bld4u(r_fs, own16, 0);
      ws_n = bld4(r_ws, own16, 0);
      if (!REGS) sp.part_n = bld4(r_part, own16, 0);
      sp.al_n = bld4(r_part, own16, 4u * (uint32_t)Hs);
      request_fix(0, 1);
    }
    part = 0.f;
    run_stream(q, [&](int j) __attribute__((always_inline)) {
      TC_STAMP(2)
      const uint32_t pj = (uint32_t)j * kPlane;
      const u4 fs = fs_n;
      const f4 ws = ws_n;
      const int fx0 = fx0_n, fx1 = fx1_n;
      for (int e = fx0; e < fx1; ++e) fold_row_pw(fix[e], vrow, aACC, Hs);
      f4 F = own_rows(vrow, 0);
      f4 al;  // alpha'_{t-1} of the owned states
      if constexpr (SPLIT) {
        F += REGS ? rget(j) : sp.part_n;  // (the first walk's row sums)
        // (the second half's planes are what LDS holds now; the first half's were parked by the first walk)
        al = j < sp.planes_a ? sp.al_n : lds4(kA0 + own16 + (uint32_t)(j - sp.planes_a < 0 ? 0 : j - sp.planes_a) * kPlane);
      } else {
        al = lds4(kA0 + own16 + pj);
      }
      const f4 bt = bt_n;
      const f4 a = f4{tied_fwd_state<GAMMA>(kPB, aGM, inv_prev, fs.x, ws.x, F.x, al.x, bt.x, gs, dpart),
                      tied_fwd_state<GAMMA>(kPB, aGM, inv_prev, fs.y, ws.y, F.y, al.y, bt.y, gs, dpart),
                      tied_fwd_state<GAMMA>(kPB, aGM, inv_prev, fs.z, ws.z, F.z, al.z, bt.z, gs, dpart),
                      tied_fwd_state<GAMMA>(kPB, aGM, inv_prev, fs.w, ws.w, F.w, al.w, bt.w, gs, dpart)};
      part += hsum(a);
      bst4_aux<TC_PW_HIST_AUX>(hist_t, own16 + pj, a);  // alpha_t: its history row (the backward pass is its next reader)
      rset(j, a);                                        // ... and the registers, for the frame's tail
      {  // the next plane's tables (index clamped: every request of the frame is unconditional)
        const int jn = j + 1 < planes ? j + 1 : j;
        fs_n = bld4u(r_fs, own16, (uint32_t)jn * kPlane);
        ws_n = bld4(r_ws, own16, (uint32_t)jn * kPlane);
        if (GAMMA) bt_n = bld4(brow, own16, (uint32_t)jn * kPlane);
        if constexpr (SPLIT) {
          if (!REGS) sp.part_n = bld4(r_part, own16, (uint32_t)jn * kPlane);
          sp.al_n = bld4(r_part, own16, (uint32_t)(jn < sp.planes_a ? jn : 0) * kPlane + 4u * (uint32_t)Hs);
        }
        request_fix(jn, SPLIT ? 1 : 0);
      }
      TC_STAMP(3)
    }, SPLIT ? 1 : 0);
    // alpha'_t = alpha_t + leaky * pi * asum_t: pi comes back from L2 while the block sum forms
    // (requests of planes the graph does not have lie beyond their descriptors and return zeros: every element of the
    // array is assigned unconditionally -- assigned under a condition, the compiler carries such an array through the
    // frame as one 28-register value and spills it)
    f4 cp[kMaxPlanes];
#pragma unroll
    for (int j = 0; j < kMaxPlanes; ++j) cp[j] = bld4(r_pi, own16, j * kPlane);
    f4 yp = mk4(0.f);
    if (GAMMA) {  // y_{t-1} for the derivative row's l2 term (this CU read the row a frame ago: L2)
      yp = row_ld(make_rsrc(p.y + ((int64_t)(t - 1) * S + s) * p.y_stride, row_bytes), own16, p.y_vec);
      block_sum2(part, dpart, aRed, wave, lane);  // its barrier also completes gamma_{t-1}
      asum = part;
    } else {
      asum = block_sum_a(part, aRed, wave, lane);  // every wave has finished its walks: the gather buffer may change
    }
    part_tot = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxPlanes; ++j)
      if (j < planes) {
        const f4 a = R[j] + (leaky * cp[j]) * asum;
        source_st(j, a);
        part_tot += hsum(a);
      }
    if (GAMMA) {
      // the derivative row of frame t-1: gamma_{t-1} * (c_t / c^_t)
      const float c = __builtin_amdgcn_rcpf(dpart);
      const float sa = p.deriv_weight * (kGammaInvScale * (c * __builtin_amdgcn_rcpf(chat)));
      if (own_pdfs()) {
        const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t - 1) * S + s) * p.deriv_stride, row_bytes);
        const u4 gu = lds4u(aGM + own16);
        lds4_st(aGM + own16, mk4(0.f));
        f4 o = sa * f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} - p.l2_scale * yp;
        if (ACCUM) o += row_ld(drow, own16, p.d_vec);
        row_st(drow, own16, p.d_vec, o);
      }
      xent_zero_row(t - 1);
      chat = c * asum * __builtin_amdgcn_rcpf(n_t);  // c^_{t+1} = c_t asum_t / n_t
    }
    if (t < T && own_pdfs()) {
      y2 += hsum(yreg * yreg);
      lds4_st(kPB + own16, exp4(yreg));
    }
    if (tid == 0) {
      asums.st(t, asum);
      if (fn) fn[t] = asum;
    }
    inv_prev = __builtin_amdgcn_rcpf(asum);
    TC_STAMP(4)
  }

  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  __device__ __forceinline__ float forward_total(double bad = 0.0) {
    const float tot = block_sum_a(part_tot, aRed + 4u * kWaves, wave, lane);
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      // [K] log-prob = log(tot) + sum over t < T of log(alpha-sum_t): the scales divided out of frames 1..T
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(asums.ld(t));
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d) + bad;  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
    return tot;
  }

  // ================================================================================================== backward
  // ---- [K] BetaDashLastFrame, Beta(T): beta'_T(h) = 1 / tot on the real states, beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h)
  // (PURE: b_T = 1, a recursion with normalisers of its own; B_T goes to the B history for the partner's gamma_{T-1})
  template <bool PURE>
  __device__ __forceinline__ void backward_begin(float b_T) {
    part = 0.f;
    for (int j = 0; j < planes; ++j) part += hsum(leaky * bld4(r_pi, own16, j * kPlane)) * b_T;
    bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);  // also orders the reuse of the gather buffer
    stream_begin(p.bwd);
    if (own_pdfs()) {
      lds4_st(kPB + own16, exp4(row_ld(make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes), own16, p.y_vec)));
      lds4_st(aGM + own16, mk4(0.f));
    }
    __syncthreads();  // exp(y_{T-1}) complete: Y_{T-1}(g) = beta_T(g) * p_{T-1}(f(g))
#pragma unroll
    for (int j = 0; j < kMaxPlanes; ++j) {
      const int h0 = 4 * ((int)tid + kThreads * j);
      const u4 fs = bld4u(r_fs, own16, j * kPlane);
      // beta_T = beta'_T + its leaky sum on the graph's states, zero on phantom positions and planes the graph does not have
      const f4 b = f4{h0 < H ? b_T + bsum : 0.f, h0 + 1 < H ? b_T + bsum : 0.f, h0 + 2 < H ? b_T + bsum : 0.f, h0 + 3 < H ? b_T + bsum : 0.f};
      R[j] = b;
      if (j < planes) {
        if (PURE) bst4_aux<0>(bhist_row(T), own16 + j * kPlane, b);  // B_T
        source_st_bwd(j, f4{b.x * ldsf(kPB + (fs.x & 0xffffu)), b.y * ldsf(kPB + (fs.y & 0xffffu)),
                            b.z * ldsf(kPB + (fs.z & 0xffffu)), b.w * ldsf(kPB + (fs.w & 0xffffu))});
      }
    }
    stamps_reset();
  }

  // the values of plane j a backward pass needs from memory: tables, alpha_t, alpha_{t+1}  (PURE: no alpha)
  template <bool PURE>
  __device__ __forceinline__ void request_bwd(int j, const rsrc_t &hist_t, const rsrc_t &hist_up, int half = 0) {
    const uint32_t pj = (uint32_t)j * kPlane;
    fs_n = bld4u(r_fs, own16, pj);
    ws_n = bld4(r_ws, own16, pj);
    cp_n = bld4(r_pi, own16, pj);
    if (!PURE) {
      al_n = bld4_aux<TC_PW_AL_AUX>(hist_t, own16, pj);
      aup_n = bld4_aux<TC_PW_AUP_AUX>(hist_up, own16, pj);
    }
    request_fix(j, half);
  }

  // frame t = T-1..0   ([K] BetaDashGeneralFrame(t) + Beta(t)); returns true after frame 0.
  // PURE (role B above the meeting frame, B'_T = 1): no gamma, normaliser n_t = sum_h U_t(h) / H, B_t to the B history
  template <bool PURE>
  __device__ __forceinline__ bool backward_frame(int t) {
    Chunk6 q[kBuffers];
    request_first(q);
    const rsrc_t hist_t = hist_row(t), hist_up = hist_row(t + 1);
    const rsrc_t r_part = part_rsrc();
    if (!SPLIT)
      request_bwd<PURE>(0, hist_t, hist_up);
    else
      request_fix(0, 0);
    float asum_t = 1.f;
    if (!PURE && asums.g) asum_t = asums.ld(t);  // (from the workspace: requested ahead of the barrier)
    __syncthreads();  // Y, exp(y_t) ready; gamma zero
    TC_STAMP(0)
    if (!PURE && !asums.g) asum_t = asums.ld(t);
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
    if constexpr (SPLIT) {
      first_walk<false>(q);  // the cells whose destination lies in the first half; the second half of Y_t is in LDS behind it
      request_bwd<PURE>(0, hist_t, hist_up, 1);
      sp.part_n = bld4(r_part, own16, 0);
    }
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f, part_u = 0.f;
    if (TC_PW_BWD_PRIO) age_prio_on(wave);
    run_stream(q, [&](int j) __attribute__((always_inline)) {
      TC_STAMP(2)
      const u4 fs = fs_n;
      const f4 ws = ws_n, cp = leaky * cp_n, aup = PURE ? mk4(0.f) : aup_n;
      const f4 al = PURE ? mk4(0.f) : al_n + cp * asum_t;  // alpha'_t of the owned states (the history keeps alpha_t)
      const f4 bo = rget(j);                               // beta_{t+1} (PURE: B_{t+1})
      const int fx0 = fx0_n, fx1 = fx1_n;
      for (int e = fx0; e < fx1; ++e) fold_row_pw(fix[e], vrow, aACC, Hs);
      f4 a = own_rows(vrow, 0);
      if constexpr (SPLIT) a += sp.part_n;
      a.x = tied_bwd_state<PURE>(kPB, aGM, fs.x, ws.x, bo.x, al.x, aup.x, 0.f, a.x, inv_as, 0.f);
      a.y = tied_bwd_state<PURE>(kPB, aGM, fs.y, ws.y, bo.y, al.y, aup.y, 0.f, a.y, inv_as, 0.f);
      a.z = tied_bwd_state<PURE>(kPB, aGM, fs.z, ws.z, bo.z, al.z, aup.z, 0.f, a.z, inv_as, 0.f);
      a.w = tied_bwd_state<PURE>(kPB, aGM, fs.w, ws.w, bo.w, al.w, aup.w, 0.f, a.w, inv_as, 0.f);
      const f4 b = PURE ? a : a * inv_as;  // [K] * inv_arbitrary_scale: beta'_t (PURE: U_t)
      part += hsum(cp * b);
      if (PURE) part_u += hsum(a);
      if (!PURE && t == 0) part_ab += hsum(al * b);
      rset(j, b);  // (beta_{t+1} of the plane has just been used)
      // the next plane's values, requested behind this plane's arithmetic (both sets at once do not fit the registers) and
      // a whole sub-stream ahead of their use; index clamped: every request of the frame is unconditional
      request_bwd<PURE>(j + 1 < planes ? j + 1 : j, hist_t, hist_up, SPLIT ? 1 : 0);
      if constexpr (SPLIT) sp.part_n = bld4(r_part, own16, (uint32_t)(j + 1 < planes ? j + 1 : j) * kPlane);
      TC_STAMP(3)
    }, SPLIT ? 1 : 0);
    if (TC_PW_BWD_PRIO) __builtin_amdgcn_s_setprio(0);
    // beta'_t and the forward pdfs again, for the Y update behind the two barriers
    // y_{t-1} (at t == 0 frame 0 again) for the next frame's exp(y), y_t once more for the derivative row's l2 term (this CU
    // read it a frame ago: L2) -- requested here rather than held in registers through the walks
    const f4 ynext = row_ld(make_rsrc(p.y + ((int64_t)(t > 0 ? t - 1 : 0) * S + s) * p.y_stride, row_bytes), own16, p.y_vec);
    const f4 ynow = row_ld(make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes), own16, p.y_vec);
    u4 fsT[kMaxPlanes];
#pragma unroll
    for (int j = 0; j < kMaxPlanes; ++j) fsT[j] = bld4u(r_fs, own16, j * kPlane);  // (unconditional: see the forward tail)
    float inv_n = 1.f;
    if (PURE) {
      // n_t = sum_h U_t(h) / H; B'_t = U_t / n_t; leaky sum of B'_t
      block_sum2(part, part_u, aRed, wave, lane);
      const float n = part_u * (1.0f / (float)H);
      inv_n = __builtin_amdgcn_rcpf(n);
      bsum = part * inv_n;
      if (tid == 0) bn[t] = __builtin_amdgcn_rcpf(inv_n);  // (the normaliser actually applied)
    } else {
      bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    }
    if (!PURE && own_pdfs()) {
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, row_bytes);
      const u4 gu = lds4u(aGM + own16);
      lds4_st(aGM + own16, mk4(0.f));
      const f4 g = f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} * kGammaInvScale;
      if (t == 0) part_g = hsum(g);
      f4 o = p.deriv_weight * g - p.l2_scale * ynow;
      if (ACCUM) o += row_ld(drow, own16, p.d_vec);
      row_st(drow, own16, p.d_vec, o);
    }
    if (!PURE) xent_zero_row(t);
    if (!PURE && t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum_a(part_ab, aRed + 4u * kWaves, wave, lane);
      const float gsum = block_sum_a(part_g, aRed + 8u * kWaves, wave, lane);
      if (tid == 0) {
        p.seq_ab[s] = ab;
        p.seq_gsum[s] = gsum;
      }
      return true;
    }
    // exp(y_{t-1}) overwrites exp(y_t) in place -- its readers are behind the reduction's barrier -- and one more barrier
    // publishes it to the Y update
    if (own_pdfs()) lds4_st(kPB + own16, exp4(ynext));
    __syncthreads();
    // beta_t = beta'_t + leaky-sum; next frame's gather source Y_{t-1} = beta_t * p_{t-1}(f)
#pragma unroll
    for (int j = 0; j < kMaxPlanes; ++j) {
      const f4 b = PURE ? R[j] * inv_n + bsum : R[j] + bsum;
      R[j] = b;  // beta_t (B_t) for the next frame's passes
      if (j < planes) {
        if (PURE) bst4_aux<0>(bhist_row(t), own16 + j * kPlane, b);  // B_t for the partner
        source_st_bwd(j, f4{b.x * ldsf(kPB + (fsT[j].x & 0xffffu)), b.y * ldsf(kPB + (fsT[j].y & 0xffffu)),
                            b.z * ldsf(kPB + (fsT[j].z & 0xffffu)), b.w * ldsf(kPB + (fsT[j].w & 0xffffu))});
      }
    }
    TC_STAMP(4)
    return false;
  }
};

template <bool ACCUM, bool WANT_DERIV, int MAXP, bool SPLIT = false>
__global__ __launch_bounds__(kThreads) void den_tied_planes_kernel(const DenParams p) {
  PlaneSeq<ACCUM, MAXP, TC_PW_BUFFERS_FUSED, SPLIT, SPLIT> q(p, (int)blockIdx.x);
  const int T = q.T;
  // ---- forward: alpha'_0, frames 1..T, total probability
  q.forward_begin();
  for (int t = 1; t <= T; ++t) q.template forward_frame<false>(t);
  q.stamps_flush(0);
  const float tot = q.forward_total();
  if (!WANT_DERIV) return;
  // ---- backward: beta'_T = 1 / tot, frames T-1..0 with gamma
  q.template backward_begin<false>(__builtin_amdgcn_rcpf(tot));
  for (int t = T - 1; t > 0; --t) q.template backward_frame<false>(t);
  q.template backward_frame<false>(0);
  q.stamps_flush(128);
}

// =========================================================================================================
// Two workgroups per sequence that meet in the middle (batches of at most half the CUs): den_tied_mitm.hip's scheme
// =========================================================================================================
// ROLE F: alpha forward over frames 1..M exactly as the fused kernel, the hand-over, then frames M+1..T with gamma_{t-1}
template <bool ACCUM, int MAXP, bool SPLIT>
__device__ __forceinline__ void planes_mitm_forward(const DenParams &p, const MitmParams &mq, int s) {
  PlaneSeq<ACCUM, MAXP, SPLIT ? TC_PW_BUFFERS_FUSED : TC_PW_BUFFERS_PAIR, SPLIT> q(p, s, 0);
  const int T = q.T, M = mq.M;
  q.forward_begin();
  for (int t = 1; t <= M; ++t) q.template forward_frame<false>(t);
  publish(mq.sync + 4 + 2 * s, q.tid);
  const bool partner_ok = await(mq.sync + 4 + 2 * s + 1, q.tid, mq.aScr + 4u);
  if (q.own_pdfs()) lds4_st(q.aGM + q.own16, mk4(0.f));  // gamma starts at zero (the block sum below publishes it)
  {
    // c_M = 1 / sum_g alpha_M(g) B_M(g);  c^_{M+1} = c_M asum_M / n_M
    const rsrc_t aM = q.hist_row(M), bM = q.bhist_row(M);
    float d = 0.f;
    for (int j = 0; j < q.planes; ++j) d += hsum(bld4(aM, q.own16, j * kPlane) * bld4(bM, q.own16, j * kPlane));
    d = block_sum_a(d, q.aRed + 4u * kWaves, q.wave, q.lane);
    q.chat = __builtin_amdgcn_rcpf(d) * q.asum * __builtin_amdgcn_rcpf(vload_f32(q.bn + M));
  }
  for (int t = M + 1; t <= T; ++t) q.template forward_frame<true>(t);
  q.forward_total(partner_ok ? 0.0 : (double)__builtin_nanf(""));
}

// ROLE B: frames T-1..M with normalisers of its own and no gamma, the hand-over, then the fused kernel's backward frame
template <bool ACCUM, int MAXP, bool SPLIT>
__device__ __forceinline__ void planes_mitm_backward(const DenParams &p, const MitmParams &mq, int s) {
  PlaneSeq<ACCUM, MAXP, SPLIT ? TC_PW_BUFFERS_FUSED : TC_PW_BUFFERS_PAIR, SPLIT> q(p, s, 1);
  const int T = q.T, M = mq.M;
  q.template backward_begin<true>(1.0f);  // B'_T = 1
  for (int t = T - 1; t >= M; --t) q.template backward_frame<true>(t);
  publish(mq.sync + 4 + 2 * s + 1, q.tid);
  const bool partner_ok = await(mq.sync + 4 + 2 * s, q.tid, mq.aScr + 4u);
  {
    // asum_0..M from role F; c_M = 1 / sum_g alpha_M(g) B_M(g); from here on beta = c_M B: Kaldi's scale
    for (int i = (int)q.tid; i <= M; i += kThreads) ldsf_st(q.aAsum + 4u * (uint32_t)i, vload_f32(q.fn + i));
    const rsrc_t aM = q.hist_row(M), bM = q.bhist_row(M);
    float d = 0.f;
    for (int j = 0; j < q.planes; ++j) d += hsum(bld4(aM, q.own16, j * kPlane) * bld4(bM, q.own16, j * kPlane));
    d = block_sum_a(d, q.aRed + 4u * kWaves, q.wave, q.lane);  // (its barrier also publishes the frame sums)
    const float c = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      q.R[j] = q.R[j] * c;  // beta_M = c_M B_M, where frame M-1 looks for beta_{t+1}
      if (j < q.planes) q.source_scale(j, c);  // Y_{M-1}
    }
  }
  for (int t = M - 1; t >= 0; --t)
    if (q.template backward_frame<false>(t)) break;
  if (!partner_ok && q.tid == 0) p.seq_ab[s] = __builtin_nanf("");
}

template <bool ACCUM, int MAXP, bool SPLIT = false>
__global__ __launch_bounds__(kThreads) void den_tied_planes_mitm_kernel(const DenParams p, const MitmParams q) {
  const uint32_t ticket = take_ticket(q);
  const int s = (int)(ticket >> 1);
  if (s >= p.S) return;
  if ((ticket & 1u) == 0u)
    planes_mitm_forward<ACCUM, MAXP, SPLIT>(p, q, s);
  else
    planes_mitm_backward<ACCUM, MAXP, SPLIT>(p, q, s);
}

}  // namespace

bool planes_mitm_fits(const DenParams &p) {
  return p.L.planewise && !p.L.asum_global && p.T >= 2 && p.deriv && p.beta_hist && p.fwd_norm && p.bwd_norm && p.mitm_sync &&
         (size_t)layout_lds_bytes(p.L, p.T) + 16u <= (size_t)kLdsLimitBytes;
}

int launch_den_tied_planes_mitm(const DenParams &p, int accumulate, hipStream_t stream) {
  if (!planes_mitm_fits(p) || p.L.PV != kPvSmall || p.L.JV < 5 || p.L.JV > kJvPlanesSplit) return TC_ERR_UNSUPPORTED;
  if (p.L.src_planes < p.L.JV && (!p.src_scratch || !p.part_scratch)) return TC_ERR_WORKSPACE;
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T) + 16u;
  MitmParams q;
  q.sync = p.mitm_sync;
  q.M = p.T / 2;
  if (const int m = debug_value(kDbgPlanesMeetAt)) q.M = std::min(std::max(m, 1), p.T - 1);  // (diagnostic: the meeting frame)
  q.aScr = (uint32_t)layout_lds_bytes(p.L, p.T);
  TC_HIP_CHECK(hipMemsetAsync(p.mitm_sync, 0, mitm_sync_bytes(p.S), stream));
  void (*k)(const DenParams, const MitmParams) = nullptr;
  if (p.L.src_planes < p.L.JV)
    k = accumulate ? den_tied_planes_mitm_kernel<true, kJvPlanesSplit, true> : den_tied_planes_mitm_kernel<false, kJvPlanesSplit, true>;
  else if (p.L.JV <= 6)
    k = accumulate ? den_tied_planes_mitm_kernel<true, 6> : den_tied_planes_mitm_kernel<false, 6>;
  else
    k = accumulate ? den_tied_planes_mitm_kernel<true, kJvPlanes> : den_tied_planes_mitm_kernel<false, kJvPlanes>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(2 * p.S), dim3(kThreads), lds, stream, p, q);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

int launch_den_tied_planes(const DenParams &p, int accumulate, hipStream_t stream) {
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (!p.L.planewise || lds > (size_t)kLdsLimitBytes || p.L.PV != kPvSmall || p.L.JV < 5 || p.L.JV > kJvPlanesSplit) return TC_ERR_UNSUPPORTED;
  void (*k)(const DenParams) = nullptr;
  if (p.L.src_planes < p.L.JV) {  // split gather source (28673..40960 positions)
    if (!p.src_scratch || !p.part_scratch) return TC_ERR_WORKSPACE;
    if (!p.deriv)
      k = den_tied_planes_kernel<false, false, kJvPlanesSplit, true>;
    else
      k = accumulate ? den_tied_planes_kernel<true, true, kJvPlanesSplit, true> : den_tied_planes_kernel<false, true, kJvPlanesSplit, true>;
  } else if (p.L.JV <= 6) {
    if (!p.deriv)
      k = den_tied_planes_kernel<false, false, 6>;
    else
      k = accumulate ? den_tied_planes_kernel<true, true, 6> : den_tied_planes_kernel<false, true, 6>;
  } else {
    if (!p.deriv)
      k = den_tied_planes_kernel<false, false, kJvPlanes>;
    else
      k = accumulate ? den_tied_planes_kernel<true, true, kJvPlanes> : den_tied_planes_kernel<false, true, kJvPlanes>;
  }
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
