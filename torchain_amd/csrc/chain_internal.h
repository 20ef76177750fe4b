// Internal declarations shared by the host code and the HIP kernels of libtorchain_hip.so.
// Public C ABI: include/torchain_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "torchain_hip.h"

namespace tc {

// ---- fused denominator kernel geometry ---------------------------------------------------------
// One workgroup per sequence, 16 wavefronts of 64 lanes (the CU's maximum), one workgroup per CU:
// the frame recursion is serial, so everything a sequence needs per frame lives in that CU's LDS.
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
// the two kernel instantiations <JV, PV>: float4s of states / pdfs owned per thread
constexpr int kJvSmall = 2048 / kThreads, kPvSmall = 1024 / kThreads, kJvLarge = 4096 / kThreads, kPvLarge = 3072 / kThreads;
constexpr int kPvMid = 2048 / kThreads;  // 4097..8192 pdfs
constexpr int kJvMid = 3072 / kThreads;  // tied graphs of 8193..12288 positions: 12 states per thread
constexpr int kMaxRowLen = 32;             // longer in/out-arc lists are split into virtual rows
constexpr int kLdsLimitBytes = 160 * 1024; // gfx950 LDS per CU / per workgroup

// One transition as the kernels stream it: probability + two 16-bit dword indices (the state to
// gather alpha'/beta from and the pdf), 8 bytes.  [K] DenominatorGraphTransition is 12 bytes.
struct ArcRec {
  float w;
  uint32_t idx;  // state | (pdf << 16)
};

// GENERAL graphs.  A "row" is the in-arc list (forward) or out-arc list (backward) of one state, at most
// kMaxRowLen long.  Rows are sorted by length and dealt 64 at a time into "slots"; lane l of a slot walks
// row l.  Every wave gets one flat, self-describing stream of 8-byte cells laid out [pair][lane][2], so a
// wave instruction loads 64 lanes x 16 contiguous bytes and the stream is prefetched with no dependent
// address computation:
//   arc cell  : {w, pdf*4 | state*4 << 16}: the two LDS byte offsets the lane gathers from, ready to
//               use (exp(y) sits at LDS offset 0, alpha'/beta at a compile-time base that the ds_read
//               immediate supplies), so address math is one AND and one shift per cell.
//   ROW cell  : {accumulator slot | state << 16 of the row that starts here, kRowFlag | dummy offsets}.
//               It commits the previous row's sum with a plain LDS store.  All 64 lanes of a wave hit
//               their ROW cells at the same stream position, so the flag test (v_readfirstlane of
//               the cell's own offset word) feeds a scalar branch.
//   padding   : arc cell with w = 0, offsets 0 (adds 0)
// A stream ends with a ROW(dummy) cell and is padded to a multiple of kStreamUnroll cells; the whole
// array ends with readable padding so the prefetch never needs a bounds check.  Every row owns its
// accumulator slot, so no LDS float atomics are needed (ds_add_f32 costs 192 cycles per
// wave-instruction on gfx950, profiles/microbench): a state whose arc list fits one row uses slot =
// state; each further chunk of a longer list gets a private slot >= Hs + 4, and the thread that owns
// the state folds those slots in before it reads the sum ("fix-up" list, sorted by owner thread).
//
// TIED graphs use owner-computes schedules instead (schedule_owner.cpp: build_owner): states are permuted so
// that the thread owning a state walks its row, 6-byte cells without ROW cells, row ends in mask words.
constexpr uint32_t kRowFlag = 1u;  // low two bits of a cell's offset word are free (offsets are multiples of 4)
// general streams: cells per loop iteration, two 8-cell chunks in ping-pong (a third buffer was measured:
// no gain, and it pushes the fused kernel past 128 VGPRs)
constexpr int kStreamUnroll = 16;
constexpr int kStreamUnrollTied = 8;  // tied streams: a wave's range is padded to whole chunks
// ... and to at least this many chunks: the tied kernel keeps a prefix of every wave's stream in registers
// (den_tied_kernel.hip), which must exist whatever the graph
constexpr int kTiedMinChunks = 6;
constexpr int kMaxIndex = 1 << 14;
// Tied graphs of 16385..28672 positions ("plane-wise" form of the owner-computes kernel, den_tied_planes.hip; kJvPlanesSplit
// below takes it to 40960): the gather
// source alone takes 4 bytes per position of the CU's 160 KB, so a thread's 20-28 states are taken one float4 ("plane")
// at a time -- walk the four rows of the plane, then its per-state pass -- and the row sums of all planes share four
// rows per wave.  Cells carry positions (16 bits), not byte offsets.
constexpr int kJvPlanes = 7;
constexpr int kMaxPlanePositions = 4096 * kJvPlanes;
// ... and of 28673..40960 positions (round 6): the gather source no longer fits LDS as a whole, so a frame is walked in TWO
// sub-passes -- the cells whose source lies in the first half of the positions (planes 0 .. ceil(planes / 2) - 1), then,
// with the second half of the source brought in from the workspace, the others; a plane's row sums of the first sub-pass
// wait in the workspace (DenParams::part_scratch) for the second, which runs the per-state pass.  ScheduleHost::halves.
constexpr int kJvPlanesSplit = 10;
constexpr int kMaxSplitPositions = 4096 * kJvPlanesSplit;

struct ScheduleHost {
  std::vector<ArcRec> cells;       // all waves' streams, [cell][lane] (final layout [pair][lane][2])
  std::vector<uint32_t> cells6;    // tied graphs: 6-byte cells, [pair][lane]{w0, w1, off0 | off1 << 16}
  std::vector<int2> wave_range;    // kWaves x {first cell, number of cells (multiple of kStreamUnroll)}
  std::vector<int32_t> fix_begin;  // kThreads + 1: range of fix-up entries owned by each thread
  std::vector<int2> fix;           // {state, extra slot}
  int32_t extra_slots = 0;         // accumulator slots beyond Hs + 4
  int64_t conflict_cost = 0, conflict_free_cost = 0, conflict_bound = 0;  // LDS cycles of the arc gathers: as placed / if conflict-free
  int64_t real_arcs = 0, padded_arcs = 0;
  int32_t rows = 0;
  // owner-computes schedules of tied graphs (schedule_owner.cpp: build_owner)
  std::vector<uint32_t> masks;        // [wave][mask_stride]: row-end bits of pairs 8 j .. 8 j + 7 and the rows ended before its two chunks (emit_owner_stream)
  int32_t mask_stride = 0;
  std::vector<int32_t> extra_first;   // per wave: index of its first secondary-row slot group
  int32_t nfix = 0;                   // number of fix-up entries (0: no barrier after the walk)
  // plane-wise form (kMaxIndex < positions <= kMaxPlanePositions): a wave's stream is `subs` sub-streams -- its secondary
  // rows, then one per plane -- each padded to whole chunks, with mask words of its own: wave_range[wave * subs + sub],
  // masks[(wave * subs + sub) * mask_stride ...]; fix_begin is [thread][plane] (+ 1)
  int32_t subs = 0;
  // 2: split gather source (kMaxPlanePositions < positions <= kMaxSplitPositions): the wave's stream is `subs` sub-streams
  // per half (wave_range[wave * halves * subs + half * subs + sub]), a cell's field is its position inside its half,
  // fix_begin is [half][thread][plane] (+ 1)
  int32_t halves = 1;
};

struct ScheduleDev {
  const ArcRec *cells;
  const int2 *wave_range;
  const int32_t *fix_begin;
  const int2 *fix;
  const uint32_t *masks = nullptr;      // owner-computes schedules only
  const int32_t *extra_first = nullptr;
  int32_t mask_stride = 0, nfix = 0;
  const void *cells_pair = nullptr;     // tied graphs of at most 8192 positions: the stream with offsets = position * 8
  int32_t subs = 0;                     // plane-wise form: sub-streams per wave and half (ScheduleHost::subs), else 0
  int32_t halves = 1;                   // ... and the number of halves of the gather source (ScheduleHost::halves)
};

// ---- graphs too large for the on-chip layout ("streamed" path, den_slab_kernel.hip) ------------------
// alpha / beta live in HBM/L2 as [slab of G sequences][state][G] matrices, G = 16 or 32 per graph (lanes run over
// sequences, one row of a list per group of G lanes of a wave).  A LIST is a set of rows (states, or pdfs) with their entries: by destination
// (forward), by source (backward beta'), and -- general graphs only -- by pdf (gamma: one group per pdf sums its
// arcs, so the derivative needs no float atomics).  Rows are taken 64 / G at a time ("bundle"), sorted by length.
constexpr int kSlabRowsPerBlock = 64;  // rows of a list per block of 256 threads: G bundles
struct SlabRow {       // 32 bytes: two 16-byte loads per lane
  int32_t row;         // state (arc lists) or pdf (by-pdf list); -1: this group of the bundle has no row
  int32_t n;           // entries of this row
  int32_t f_off, s_off;  // tied graphs: 4 G * the state's forward / special self-loop pdf (byte offset of its row), -1: none
  float ws, pi;        // tied graphs: self-loop probability; every state list: initial probability
  float K;             // tied graphs, by destination: sum over the in-arcs of w * pi(src)
  float pad;
};
// entries: W dwords per step (tied arc lists {4 G * other state, w}: W = 2; general lists {4 G * a, 4 G * b, w, pi}:
// W = 4; by destination {src, pdf, w, pi(src)}, by source {dst, pdf, w, 1}, by pdf {src, dst, w, pi(src)}), a bundle's
// steps in chunks of 16 / W: dword c of step i of row q at [chunk][16 q + W i + c], 64 / G rows per chunk.  Steps past
// a row's end: zeros.
struct SlabListDev {
  const SlabRow *rows = nullptr;   // [bundles][64 / G]
  const int2 *head = nullptr;      // [bundles] {first chunk, steps}
  const uint32_t *rec = nullptr;   // [chunks + 3][64 / G][16]
  int32_t bundles = 0;
};
struct SlabListHost {
  int W = 2, G = 16;
  std::vector<SlabRow> rows;
  std::vector<int32_t> head;       // 2 per bundle
  std::vector<uint32_t> rec;
  int32_t bundles = 0;
};

struct BigDev {
  SlabListDev in, out, pdf;
  int G = 16;            // sequences per slab
  int in_blocks = 0, out_blocks = 0;  // blocks (of G bundles) of the two state lists; hb = the larger: rows of the per-block partials
  int hb = 0;
  // tied graphs (work-graph states): the arc lists hold the non-special arcs only and exp(y) is applied
  // per state, so an arc costs ONE row gather per pass; gamma comes from per-state quantities, added by the
  // backward kernel to fixed-point accumulators
  int tied = 0;
  const int32_t *f_off = nullptr;  // tied: per state, 4 G * forward pdf (-1: none)
};

struct DenGraphDev {
  BigDev big;
  ScheduleDev fwd, bwd;
  const float *pi;  // initial probs padded to Hs
  const uint32_t *tied_fs = nullptr;  // tied graphs: per state, forward-pdf*4 | self-loop-pdf*4 << 16
  const float *tied_w = nullptr;      // tied graphs: per state, self-loop probability (0 if none)
  void *blob = nullptr;
  // Two-sequence kernel (den_tied_pair.hip) or fused kernel for batches beyond one sequence per two CUs: decided per
  // graph and device by timing both once (api.cpp: tune_den_variant).  -1 not decided yet, -2 being decided.
  int pair_choice = -1;
  float tune_ms[2] = {0.f, 0.f};  // what the decision was made on: fused, two-sequence (ms for kTuneFrames frames)
};

// LDS layout of the fused kernel, in floats: [P | A | ACC | GAMMA | ALPHA? | red | asum].  exp(y)
// (P) starts at offset 0 and its region has the compile-time size PV * 4096 of the kernel
// instantiation, so the gather source A (alpha' forward, beta backward) has a compile-time base.
struct DenLayout {
  int Hs, Ps;           // H, P rounded up to a multiple of 4
  int off_a, off_acc, off_g, off_al, off_red, off_asum, total_floats;
  bool alpha_in_lds;
  int off_p2;           // tied graphs, backward: second exp(y) buffer (frame t-1 while frame t is in use)
  int JV, PV;           // template instantiation: float4s of states / pdfs owned per thread
  int acc_floats;       // size of the accumulator region: Hs + 4 + extra slots, rounded to 4
  bool planewise = false;  // den_tied_planes.hip: [P | A | 4 rows per wave + 4 + extra slots | GAMMA | red | asum]
  // the frame sums asum_0..T live in the workspace (DenParams::asum_g), not behind off_asum: utterances too long for the
  // LDS the graph leaves (plane-wise kernel, general owner-computes kernel: the kernels they replaced ran any T)
  bool asum_global = false;
  // plane-wise form: planes of the gather source that are in LDS at a time -- JV, or ceil(JV / 2) for graphs beyond
  // kMaxPlanePositions (the gather region is 4096 * src_planes floats)
  int src_planes = 0;
};

struct DenParams {
  ScheduleDev fwd, bwd;
  const float *pi;
  const float *y;
  int64_t y_stride;
  float *deriv;
  int64_t deriv_stride;
  float *alpha_hist;    // [(T+1)][S][Hs]
  float *asum_g = nullptr;  // [S][asum_stride(T)]: the frame sums when the layout says asum_global
  // split gather source (DenLayout::src_planes < JV): the second half of the running frame's gather source
  // [S][4096 * (JV - src_planes)] and the first sub-pass's row sums [S][Hs]
  float *src_scratch = nullptr, *part_scratch = nullptr;
  double *seq_logprob;  // [S]
  double *seq_y2;       // [S] sum of y^2 (for the l2 term)
  float *seq_ab;        // [S] sum_h alpha'_0 beta'_0
  float *seq_gsum;      // [S] sum_pdf gamma_0
  int S, T, H, P;
  float leaky, deriv_weight, l2_scale;
  int y_vec, d_vec;     // rows 16-byte aligned -> float4 path
  // The cross-entropy output's derivative is zero outside the numerator's posteriors: kernels that write every
  // derivative row anyway (den_tied_frames.h) also write its zero rows, so that the objective call needs no memset of
  // the whole matrix in front of them (629 MB at C3: 0.13 ms).  Null: the caller has zeroed it.
  float *xent_zero = nullptr;
  int64_t xent_stride = 0;
  int x_vec = 0;
  DenLayout L;
  const uint32_t *tied_fs;  // non-null selects the tied-graph kernel (see tc_den_graph below)
  const float *tied_w;
  int gen_owner = 0;        // general graph on owner-computes schedules: den_general_owner.hip
  BigDev big;           // streamed path only
  float *big_expy;      // [slab][P][16]  exp(y_t) of the current frame, transposed; or [T][slab][P][16], every frame (big_exp_stride)
  int64_t big_exp_stride = 0;  // floats between consecutive frames of big_expy (0: one frame at a time)
  float *big_beta;      // [2][slab][H][16]
  float *big_y;         // [slab][H][16]  tied graphs: Y = beta_{t+1} * p_t(f), the backward gather source
  uint32_t *big_gam = nullptr;  // [slab][P][G]  tied graphs: the frame's gamma, unsigned fixed point (den_device.h: kGammaScale)
  float *big_small;     // per-sequence sums and per-block partials (den_slab_kernel.hip: BigSmall)
  int big_Sp;           // sequences rounded up to a multiple of the slab width (big.G)
  float big_sum_pi;     // sum of the initial probabilities
  long long *stamps;    // diagnostic builds only (-DTC_PHASE_STAMPS): per-phase cycle totals, else unused
  // Two-CU form for small batches (den_tied_split.hip): the forward recursion writes its per-frame sums, the
  // backward recursion -- which does not need alpha -- runs at the same time on another CU with normalisers of
  // its own, and a third pass forms gamma from the two histories.  Null on the fused path.
  float *beta_hist = nullptr;   // [(T+1)][S][Hs]  B_t (row 0: B'_0), see den_tied_split.hip
  float *fwd_norm = nullptr;    // [S][T+2]        asum_0 .. asum_T, tot
  float *bwd_norm = nullptr;    // [S][T+1]        n_0 .. n_{T-1}
  // Two-sequence form (den_tied_pair.hip): normalisers of both roles [2][S][pair_norm_stride(T)], ticket + flags
  float *pair_norm = nullptr;
  uint32_t *pair_sync = nullptr;
  long long *pair_stamps = nullptr;  // diagnostic builds (-DTC_PAIR_STAMPS): [role][wave][T + 2][8] raw cycle stamps of pair 0
  int pair_extra_slots = 0;     // secondary-row slots of the graph's schedules (LDS layout of the pair kernel)
  int pair_choice = 0;        // the graph's tuned choice (DenGraphDev::pair_choice > 0)
  uint32_t *mitm_sync = nullptr;  // den_tied_mitm.hip: ticket + hand-over words (workspace of batches the two-CU forms may take)
};

}  // namespace tc

struct tc_den_graph {
  int32_t H = 0, P = 0;
  int64_t A = 0;
  std::vector<int32_t> arc_src, arc_dst, arc_pdf;
  std::vector<float> arc_prob;
  std::vector<float> initial_probs;
  tc::ScheduleHost fwd, bwd;
  // "Tied" graph: every non-self-loop arc entering a state carries the same pdf and a state has at most
  // one self-loop (chain topology after Kaldi's self-loop reordering: forward-pdf on entry, self-loop
  // pdf on the loop).  Then exp(y) factors out of the arc sums: the schedules hold only the
  // non-self-loop arcs, the forward walk gathers alpha' alone, the backward walk gathers
  // Y(g) = beta(g) * p(f(g)) alone, and the self-loops are applied per state by the owning thread.
  bool tied = false;
  // GENERAL graph on the owner-computes schedules (den_general_owner.hip, round 5): states addressed by layout position as
  // for tied graphs, 8-byte cells with the arc's pdf, no per-state tables
  bool gen_owner = false;
  std::vector<uint32_t> tied_fs;      // position order once build_owner has run
  std::vector<float> tied_w;
  std::vector<uint32_t> tied_fs_state;  // ... and in work-state order, as detect_tied left them (build_owner starts from these)
  std::vector<float> tied_w_state;
  // The graph the tied schedules are built from: the FST itself, or -- when a few states are entered
  // through arcs of more than one pdf -- its "tied-ified" version in which such a state is split into one
  // copy per entering pdf (schedule_owner.cpp: make_work_graph).  copy_first[h] .. copy_first[h+1] are the work
  // states of FST state h; work_H == H and copy_first[h] == h when nothing was split.
  int32_t work_H = 0;
  std::vector<int32_t> work_src, work_dst, work_pdf, copy_first;
  std::vector<float> work_prob, work_pi;
  std::vector<int32_t> pos;           // tied graphs: WORK state -> LDS position (see build_owner)
  std::vector<float> pi_pos;          // initial probs in position order
  tc::DenLayout layout;
  bool layout_ok = false;
  // streamed path: chosen when neither on-chip layout fits (or TC_FORCE_BIG is set)
  bool big = false;
  tc::SlabListHost big_in, big_out, big_pdf;
  int big_hb = 0;  // blocks of the longer state list (tc::BigDev::hb)
  int big_G = 16;  // sequences per slab: 32 when a slab's slice of the state matrix fits an XCD's L2 (den_graph.cpp)
  std::vector<int32_t> big_f_off;
  std::vector<int32_t> tied_f, tied_s;  // per work state: forward / special self-loop pdf, -1 if none
  float big_sum_pi = 0.f;
  std::mutex mu;
  std::map<int, tc::DenGraphDev> dev;
  std::map<int, int> preset_variant;  // device -> kernel choice fixed by the caller (tc_den_graph_set_variant)
};

// ---- numerator -----------------------------------------------------------------------------------
namespace tc {

// Per-sequence acceptors flattened into arrays.  Local state 0 of a sequence is its (possibly
// virtual) start state; states are in time order; arcs are sorted by source state.
struct NumTables {
  std::vector<int32_t> seq_state_off, seq_arc_off, seq_uniq_off;  // S + 1 each
  std::vector<int32_t> level_begin;   // [S][T + 2] local state index where each time level begins
  std::vector<int32_t> out_begin;     // per state (+1 per sequence): arc range, local arc ids
  std::vector<int32_t> in_begin;      // per state (+1 per sequence): range into in_arc
  std::vector<int32_t> in_arc;        // per arc: local arc id, grouped by destination state
  std::vector<int32_t> arc_src, arc_dst, arc_uniq;  // local ids
  std::vector<float> arc_logw;        // -arc.weight
  std::vector<float> final_logw;      // per state: -final weight, -inf when not final
  std::vector<int32_t> uniq_t, uniq_pdf;  // per unique (frame, pdf) of a sequence
  std::vector<int32_t> uniq_begin;    // per uniq (+1 per sequence): range into uniq_arc
  std::vector<int32_t> uniq_arc;      // local arc ids in increasing order
  int32_t max_states = 0, max_arcs = 0, max_uniq = 0;
};

// One slot of the per-device supervision pool (supervision.cpp): device blob, pinned staging of the same size,
// `ready` = the upload has landed, `done` = the last launch that reads the blob has finished.
struct PoolSlot {
  char *blob = nullptr, *host = nullptr;
  size_t cap = 0;
  hipEvent_t ready = nullptr, done = nullptr;
};

struct NumDev {
  const int32_t *seq_state_off, *seq_arc_off, *seq_uniq_off, *level_begin, *out_begin, *in_begin, *in_arc,
      *arc_src, *arc_dst, *arc_uniq, *uniq_t, *uniq_pdf, *uniq_begin, *uniq_arc;
  const float *arc_logw, *final_logw;
  float *stage = nullptr;  // [unique (frame, pdf) entries of all sequences]: posteriors in transit (launch_num_scatter)
  PoolSlot *slot = nullptr;
  size_t upload_bytes = 0;  // of the slot's pinned staging, to be copied to the device
  bool uploaded = false;    // the copy has been enqueued (tc_supervision_prepare); staged only: tc_supervision_stage
};

struct NumParams {
  NumDev t;
  const float *y;
  int64_t y_stride;
  float *deriv;
  int64_t deriv_stride;
  float *xent;
  int64_t xent_stride;
  double *seq_logprob;  // [S]
  int S, T, P;
  float weight;
  float deriv_scale = 1.f, xent_scale = 1.f;  // tc_chain_objf_and_grad: -1 and -xent_regularize (else 1, 1)
  // non-zero: the kernel leaves weight * posterior in t.stage instead of adding it to deriv / writing xent, so that
  // it can run beside the denominator (which writes every element of deriv); launch_num_scatter finishes the job
  int staged = 0;
  // Kaldi's cross-entropy objective sum(xent_output * xent_deriv) has as many non-zero terms as xent_deriv has: the
  // kernel that writes xent_deriv's entries multiplies each by its xent_output element and leaves the sequence's sum
  // (double) in seq_xent -- instead of a pass over both dense matrices (1.26 GB at C3).  Null: not wanted.
  const float *xent_out = nullptr;
  int64_t xent_out_stride = 0;
  double *seq_xent = nullptr;  // [S]
  // non-zero: xent / xent_out are the caller's (B, C, T) tensors, element (sequence, pdf, frame) -- the numerator touches
  // only its posteriors' entries, so the 3-D call needs no frame-major copy of either (tc_chain_step)
  int xent_bct = 0, xent_out_bct = 0, y_bct = 0;  // (y is read at the supervision's (frame, pdf) pairs only)
  int lds_states, lds_arcs, lds_uniq;
};

}  // namespace tc

struct tc_supervision {
  float weight = 1.f;
  int32_t S = 0, T = 0, P = 0;
  tc::NumTables tab;
  std::vector<size_t> blob_off;    // offsets of the tables inside a pool slot
  std::mutex mu;
  std::map<int, tc::NumDev> dev;
};

namespace tc {

int build_schedules(tc_den_graph *g);                                                  // den_graph.cpp
void build_general(tc_den_graph *g);                                                   // schedule_general.cpp
bool detect_tied(tc_den_graph *g, std::vector<char> *special);                         // schedule_owner.cpp
bool make_work_graph(tc_den_graph *g);                                                 // schedule_owner.cpp
// (count_only: fills fwd / bwd.padded_arcs and the layout alone -- what a candidate row cut would cost)
// (general: owner-computes schedules of a graph that is not chain-structured -- every arc, its pdf in the cell)
bool build_owner(tc_den_graph *g, const std::vector<char> &special, int max_row, bool count_only = false, bool general = false);  // schedule_owner.cpp
int arrange_half(const std::vector<std::vector<int64_t>> &lane_arcs, int steps, const int32_t *other,
                 const int32_t *pdf, std::vector<std::vector<int>> *pos_out);          // den_layout.cpp
// ... one gather per cell (tied schedules): step by step, a matching of lanes to banks that keeps the half-slot on its
// lower bound max(steps, most loaded bank).  pad_bank[l][k]: the bank a padding cell should gather from.
int arrange_half_matching(const std::vector<std::vector<int64_t>> &lane_arcs, int steps, const int32_t *other,
                          std::vector<std::vector<int>> *pos_out, std::vector<std::vector<int>> *pad_bank,
                          int *lower_bound);                                           // den_layout.cpp
inline int round4(int x) { return (x + 3) & ~3; }
// need_alpha (the general owner-computes kernel): only layouts with alpha' in LDS; when the frame sums of T_hint frames
// do not fit behind one, they go to the workspace (asum_global) instead of alpha' going to the history
bool compute_layout(int H, int P, int T_hint, int extra_slots, bool tied, DenLayout *L, bool need_alpha = false);
constexpr int asum_stride(int T) { return (T + 2 + 3) & ~3; }
bool compute_layout_planes(int Npos, int P, int T_hint, int extra_slots, DenLayout *L);
int64_t layout_lds_bytes(const DenLayout &L, int T);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size class) instead of before every
// launch: it is a driver call on the launch path of the hot loop.
hipError_t allow_dynamic_lds(const void *kernel, size_t lds_bytes);                    // den_graph.cpp
int launch_den(const DenParams &p, hipStream_t stream);
int launch_den_big(const DenParams &p, int accumulate, hipStream_t stream);
int launch_den_tied(const DenParams &p, int accumulate, hipStream_t stream);  // den_tied_kernel.hip
int launch_den_general_owner(const DenParams &p, int accumulate, hipStream_t stream);  // den_general_owner.hip
int launch_den_tied_planes(const DenParams &p, int accumulate, hipStream_t stream);  // den_tied_planes.hip
// ... two workgroups per sequence meeting in the middle (batches of at most half the CUs)
bool planes_mitm_fits(const DenParams &p);
int launch_den_tied_planes_mitm(const DenParams &p, int accumulate, hipStream_t stream);
// den_tied_split.hip: the backward recursion alone (to run beside a forward-only launch_den_tied), and the pass
// that forms gamma / the derivative from the two histories
int launch_den_tied_backward_only(const DenParams &p, hipStream_t stream);
int launch_den_tied_combine(const DenParams &p, int accumulate, int num_cus, hipStream_t stream);
bool split_bwd_fits(const DenLayout &L, int T);  // the backward-only kernel's and the combining pass's LDS fit one CU
constexpr int kSplitMaxSeq = 128;  // two CUs per sequence: batches of at most half the chip's CUs
// den_tied_pair.hip: two sequences per workgroup, the two recursions of a pair on two CUs meeting in the middle
int launch_den_tied_pair(const DenParams &p, int extra_slots, int accumulate, hipStream_t stream);
bool pair_fits(const DenLayout &L, int extra_slots, int T);
int tune_den_variant(tc_den_graph *g, int device);  // api.cpp
// tuning_cache.cpp: the measured choices of earlier runs, keyed by graph hash and device name
std::string tuning_cache_key(uint64_t graph_hash, const char *device_name);
bool tuning_cache_get(const std::string &key, int *two_sequence_kernel);
void tuning_cache_put(const std::string &key, int two_sequence_kernel, float fused_ms, float two_sequence_ms);
// den_tied_mitm.hip: two CUs per sequence meeting in the middle (batches of at most half the CUs)
int launch_den_tied_mitm(const DenParams &p, uint32_t *sync, int accumulate, hipStream_t stream);
bool mitm_fits(const DenLayout &L, int T);
size_t mitm_sync_bytes(int S);
int pair_norm_stride(int T);
size_t pair_sync_bytes(int S);
inline size_t pair_stamp_bytes(int T) { return ((size_t)2 * kWaves * (T + 2) * 8 * 8 + 255) & ~(size_t)255; }
int64_t big_small_floats(int hb, int P, int T, int Sp);
int launch_num(const NumParams &p, hipStream_t stream);
int launch_num_scatter(const NumParams &p, hipStream_t stream);
// Side streams and fork / join events per (device, caller stream) (made on first use; the hot path only records and waits):
// `den_side` carries the backward recursion of the two-CU form (den_tied_split.hip), `num_side` the numerator when
// the denominator leaves CUs idle (api.cpp).
struct SideStreams {
  hipStream_t den_side = nullptr, num_side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, num_fork = nullptr, num_join = nullptr;
  int num_cus = 0;
  std::recursive_mutex enqueue;  // one caller at a time records / waits on the events (a wait binds to the latest record)
};
int side_streams(hipStream_t stream, SideStreams **out);  // den_graph.cpp
void side_streams_forget(hipStream_t stream);              // ... before a caller stream is destroyed
// CUs a denominator launch of S sequences occupies (two per sequence in the two-CU form, all for the streamed path)
int den_cus_used(const DenParams &p, int num_cus);
bool den_zeroes_xent(const DenParams &p, int num_cus);  // whether the kernel launch_den_mode will take honours xent_zero
int launch_finalize(const double *den_lp, const double *num_lp, const double *y2, const float *ab, const float *gs,
                    int S, int T, float sup_weight, float l2, int have_deriv, float *results, int32_t *fail_flag,
                    hipStream_t stream, const double *xent_lp = nullptr, double *xent_total = nullptr,
                    float *loss_out = nullptr);
int launch_zero_on_fail(const int32_t *fail_flag, float *a, int64_t a_stride, float *b, int64_t b_stride,
                        const float *y, int64_t y_stride, float l2_scale, int64_t rows, int cols, hipStream_t stream);
int launch_den_reduce(const double *den_lp, const float *ab, const float *gs, int S, double *logprob_out,
                      int32_t *status_out, hipStream_t stream);
int launch_sum_double(const double *in, int n, double scale, double *out, hipStream_t stream);
int launch_xent_total(const double *in, int n, const int32_t *fail_flag, double *out, hipStream_t stream);
int launch_step_loss(const float *results3, float *loss1, hipStream_t stream);
int64_t trace_workspace_bytes();
int launch_trace_mat_mat(const float *a, int64_t a_stride, const float *b, int64_t b_stride, int64_t rows, int cols,
                         double *partial, double *out, hipStream_t stream);
int launch_layout(bool to2d, const float *in, float *out, int B, int Cn, int T, int64_t stride2d, float scale,
                  hipStream_t stream);

extern thread_local int g_last_hip_error;

// Diagnostic switches (tc_debug_set in the public header): process-wide, read when a graph is built.  They
// replace what used to be environment variables of the shipping library.
// bumped whenever a round changes a kernel the per-graph choice is timed on (tuning_cache.cpp: part of the cache key)
constexpr int kKernelGeneration = 6;
enum DebugFlag { kDbgForceGeneral = 0, kDbgForceStreamed, kDbgNoSplit, kDbgNoPdfBanks, kDbgNoBankSearch, kDbgSchedTrace, kDbgNoPhaseSplit, kDbgNoNumOverlap, kDbgNoPair, kDbgForcePair, kDbgNoTune, kDbgNoMitm, kDbgForceMitm, kDbgSlabWide, kDbgSlabNarrow, kDbgExpPerFrame, kDbgOldArrange, kDbgNoPlanes, kDbgOldGeneral, kDbgPhantomPdf0, kDbgNoPdfSearch, kDbgNoSplitSrc, kDbgPlanesMeetAt, kDbgCount };
bool debug_flag(DebugFlag f);
int debug_value(DebugFlag f);  // the switch's integer value (0: not set)

int pool_acquire(int device, size_t bytes, PoolSlot **out);                      // supervision.cpp
void pool_release(int device, PoolSlot *slot);
int64_t pool_counter(int which);  // 0: device allocations made by the pool, 1: slots reused
// launches enqueued so far, by kind (tc_debug_counter): the evaluation-step tests read them to see that a forward-only
// call enqueued no backward recursion
enum LaunchCounter { kCntDen = 0, kCntDenBackward, kCntNum, kCntNumBackward, kCntLayout, kCntDenAsumGlobal, kCntCount };
void count_launch(LaunchCounter c);
int supervision_mark_use(tc_supervision *sup, int device, hipStream_t stream);
#define TC_HIP_CHECK(expr)                        \
  do {                                            \
    hipError_t e__ = (expr);                      \
    if (e__ != hipSuccess) {                      \
      tc::g_last_hip_error = (int)e__;            \
      return TC_ERR_HIP;                          \
    }                                             \
  } while (0)

}  // namespace tc
