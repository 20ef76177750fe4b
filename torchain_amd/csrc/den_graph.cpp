// Host side of the denominator graph: construction from an FST (what the reference does by calling
// kaldi::chain::DenominatorGraph at src/my_lib_example.cpp:129-134), the choice of kernel family and its
// tables (schedule_owner.cpp, schedule_general.cpp, streamed CSR lists here), the OpenFst binary reader,
// the host replay of the schedules and the per-device immutable copies.
#include <algorithm>
#include <atomic>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {

thread_local int g_last_hip_error = 0;

hipError_t allow_dynamic_lds(const void *kernel, size_t lds_bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> granted;  // largest size already granted
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = granted[std::make_pair(kernel, device)];
  if (have >= lds_bytes) return hipSuccess;
  // the request itself, not the chip's limit: a kernel with static LDS of its own cannot be granted all of it
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e == hipSuccess) have = lds_bytes;
  return e;
}

// One set of side streams and fork / join events per (device, CALLER stream): callers on different streams neither
// share events (a wait binds to the latest record) nor serialise on one mutex.  Created on first use, all or nothing.
// (process lifetime, never destroyed: see the pools of supervision.cpp / egs_reader.cpp)
static std::mutex &g_side_mu = *new std::mutex();
static std::map<std::pair<int, hipStream_t>, SideStreams> &g_side_ctx = *new std::map<std::pair<int, hipStream_t>, SideStreams>();

// A caller stream that is about to be destroyed (api.cpp: the tuning launches' temporary stream) takes its entry along:
// the stream's value may be handed out again by the runtime, and the entry holds two streams and four events.
void side_streams_forget(hipStream_t stream) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_side_mu);
  auto it = g_side_ctx.find(std::make_pair(device, stream));
  if (it == g_side_ctx.end()) return;
  SideStreams &c = it->second;
  for (hipStream_t x : {c.den_side, c.num_side})
    if (x) {
      (void)hipStreamSynchronize(x);
      (void)hipStreamDestroy(x);
    }
  for (hipEvent_t x : {c.fork, c.join, c.num_fork, c.num_join})
    if (x) (void)hipEventDestroy(x);
  g_side_ctx.erase(it);
}

int side_streams(hipStream_t stream, SideStreams **out) {
  std::mutex &mu = g_side_mu;
  auto &ctx = g_side_ctx;
  int device = 0;
  TC_HIP_CHECK(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mu);
  SideStreams &c = ctx[std::make_pair(device, stream)];
  if (!c.den_side) {
    int num_cus = 0;
    TC_HIP_CHECK(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, device));
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    if (e != hipSuccess) {  // nothing half-built stays behind: the next call starts again
      for (hipEvent_t x : ev)
        if (x) (void)hipEventDestroy(x);
      for (hipStream_t x : st)
        if (x) (void)hipStreamDestroy(x);
      g_last_hip_error = (int)e;
      return TC_ERR_HIP;
    }
    c.fork = ev[0];
    c.join = ev[1];
    c.num_fork = ev[2];
    c.num_join = ev[3];
    c.num_side = st[0];
    c.den_side = st[1];
    c.num_cus = num_cus;
  }
  *out = &c;
  return TC_OK;
}

static std::atomic<int> g_debug[kDbgCount];
bool debug_flag(DebugFlag f) { return g_debug[f].load(std::memory_order_relaxed) != 0; }
int debug_value(DebugFlag f) { return g_debug[f].load(std::memory_order_relaxed); }
static std::atomic<int64_t> g_launches[kCntCount];
void count_launch(LaunchCounter c) { g_launches[c].fetch_add(1, std::memory_order_relaxed); }

// ---- lists of the streamed path (chain_internal.h: SlabListHost) -------------------------------------------
struct SlabEntry {
  uint32_t d[4];
};
static uint32_t f2u(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

// entries[r]: row r's entries in list order; meta[r]: its SlabRow (n is filled in here); bundle_rows: 64 / G row indices
// per bundle, -1 for none.  A bundle {r, -2, -2, ...} holds the ONE long row r ("hub"): its entries are dealt round-robin
// to the bundle's groups (entry i: group i mod NG, step i / NG), the kernels add the groups' sums (groups_sum) and only
// group 0 carries the row -- a row of several hundred entries (the popular states of a phone-LM graph) then takes
// 1 / NG of the steps, and it is a frame's longest dependent chain.
static void make_slab_list(int W, int G, const std::vector<std::vector<SlabEntry>> &entries, const std::vector<SlabRow> &meta,
                           const std::vector<int32_t> &bundle_rows, SlabListHost *out) {
  const int spc = 16 / W, NG = 64 / G, CH = NG * 16;
  out->W = W;
  out->G = G;
  out->bundles = (int32_t)(bundle_rows.size() / NG);
  out->rows.assign(bundle_rows.size(), SlabRow{-1, 0, -1, -1, 0.f, 0.f, 0.f, 0.f});
  out->head.assign((size_t)out->bundles * 2, 0);
  out->rec.clear();
  size_t chunk = 0;
  for (int32_t b = 0; b < out->bundles; ++b) {
    const int32_t *br = &bundle_rows[(size_t)b * NG];
    const bool shared = NG > 1 && br[1] == -2;
    int steps = 0;
    if (shared) {
      steps = ((int)entries[br[0]].size() + NG - 1) / NG;
    } else {
      for (int q = 0; q < NG; ++q)
        if (br[q] >= 0) steps = std::max(steps, (int)entries[br[q]].size());
    }
    const int chunks = (steps + spc - 1) / spc;
    out->head[(size_t)b * 2] = (int32_t)chunk;
    out->head[(size_t)b * 2 + 1] = shared ? (int32_t)((uint32_t)steps | 0x80000000u) : steps;
    out->rec.resize((chunk + chunks) * CH, 0u);
    auto put = [&](int q, int i, const SlabEntry &e) {
      for (int c = 0; c < W; ++c) out->rec[(chunk + i / spc) * CH + q * 16 + (i % spc) * W + c] = e.d[c];
    };
    if (shared) {
      const int32_t r = br[0];
      const int n = (int)entries[r].size();
      for (int q = 0; q < NG; ++q) {
        SlabRow m = q == 0 ? meta[r] : SlabRow{-1, 0, -1, -1, 0.f, 0.f, 0.f, 0.f};
        m.n = (n - q + NG - 1) / NG;
        out->rows[(size_t)b * NG + q] = m;
      }
      for (int i = 0; i < n; ++i) put(i % NG, i / NG, entries[r][i]);
    } else {
      for (int q = 0; q < NG; ++q) {
        const int32_t r = br[q];
        if (r < 0) continue;
        SlabRow m = meta[r];
        m.n = (int32_t)entries[r].size();
        out->rows[(size_t)b * NG + q] = m;
        for (int i = 0; i < m.n; ++i) put(q, i, entries[r][i]);
      }
    }
    chunk += chunks;
  }
  out->rec.resize((chunk + 3) * CH, 0u);  // spare chunks: the walk requests a pair of chunks ahead
}

// rows sorted by length (longest first, stable), 64 / G at a time; rows of kSlabHubLen entries or more get a bundle of
// their own
constexpr size_t kSlabHubLen = 48;
static std::vector<int32_t> bundles_by_length(const std::vector<std::vector<SlabEntry>> &entries, int G) {
  const int NG = 64 / G;
  std::vector<int32_t> order(entries.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return entries[a].size() > entries[b].size(); });
  std::vector<int32_t> out;
  size_t i = 0;
  for (; NG > 1 && i < order.size() && entries[order[i]].size() >= kSlabHubLen; ++i) {
    out.push_back(order[i]);
    for (int q = 1; q < NG; ++q) out.push_back(-2);
  }
  for (; i < order.size(); ++i) out.push_back(order[i]);
  while (out.size() % NG) out.push_back(-1);
  if (out.empty()) out.assign(NG, -1);
  return out;
}

// by-pdf list: a block of the gamma kernel is a tile of 64 consecutive pdfs; its bundles are sorted inside the tile
static std::vector<int32_t> bundles_by_tile(const std::vector<std::vector<SlabEntry>> &entries, int P) {
  std::vector<int32_t> out;
  for (int p0 = 0; p0 < P; p0 += 64) {
    std::vector<int32_t> tile(64, -1);
    for (int i = 0; i < 64 && p0 + i < P; ++i) tile[i] = p0 + i;
    std::stable_sort(tile.begin(), tile.end(), [&](int32_t a, int32_t b) {
      const size_t na = a >= 0 ? entries[a].size() + 1 : 0, nb = b >= 0 ? entries[b].size() + 1 : 0;
      return na > nb;
    });
    out.insert(out.end(), tile.begin(), tile.end());
  }
  return out;
}

static void build_big(tc_den_graph *g) {
  const int H = g->H, P = g->P, G = g->big_G;
  const uint32_t rb = 4u * G;  // bytes per row
  const int64_t A = g->A;
  std::vector<std::vector<SlabEntry>> in(H), outl(H), pdf(P);
  for (int64_t a = 0; a < A; ++a) {  // FST arc order kept
    const uint32_t s = (uint32_t)g->arc_src[a] * rb, d = (uint32_t)g->arc_dst[a] * rb, pd = (uint32_t)g->arc_pdf[a] * rb;
    const uint32_t w = f2u(g->arc_prob[a]), pis = f2u(g->initial_probs[g->arc_src[a]]);
    in[g->arc_dst[a]].push_back(SlabEntry{{s, pd, w, pis}});
    outl[g->arc_src[a]].push_back(SlabEntry{{d, pd, w, f2u(1.0f)}});
    pdf[g->arc_pdf[a]].push_back(SlabEntry{{s, d, w, pis}});
  }
  std::vector<SlabRow> ms(H), mp(P);
  for (int h = 0; h < H; ++h) ms[h] = SlabRow{h, 0, -1, -1, 0.f, g->initial_probs[h], 0.f, 0.f};
  for (int i = 0; i < P; ++i) mp[i] = SlabRow{i, 0, -1, -1, 0.f, 0.f, 0.f, 0.f};
  make_slab_list(4, G, in, ms, bundles_by_length(in, G), &g->big_in);
  make_slab_list(4, G, outl, ms, bundles_by_length(outl, G), &g->big_out);
  make_slab_list(4, G, pdf, mp, bundles_by_tile(pdf, P), &g->big_pdf);
  g->big_f_off.assign(1, -1);
  float sum_pi = 0.f;
  for (int h = 0; h < H; ++h) sum_pi += g->initial_probs[h];
  g->big_sum_pi = sum_pi;
}

// Streamed path of a tied (work) graph: arc lists without the special self-loops, per-state pdfs and self-loop
// probabilities in the rows (chain_internal.h: BigDev).
static void build_big_tied(tc_den_graph *g, const std::vector<char> &special) {
  const int H = g->work_H, G = g->big_G;
  const uint32_t rb = 4u * G;  // bytes per row
  const int64_t A = (int64_t)g->work_src.size();
  std::vector<std::vector<SlabEntry>> in(H), outl(H);
  std::vector<float> K(H, 0.f);
  for (int64_t a = 0; a < A; ++a) {
    if (special[a]) continue;
    const int32_t s = g->work_src[a], d = g->work_dst[a];
    in[d].push_back(SlabEntry{{(uint32_t)s * rb, f2u(g->work_prob[a]), 0u, 0u}});
    outl[s].push_back(SlabEntry{{(uint32_t)d * rb, f2u(g->work_prob[a]), 0u, 0u}});
    K[d] += g->work_prob[a] * g->work_pi[s];
  }
  std::vector<SlabRow> mi(H), mo(H);
  g->big_f_off.assign(H, -1);
  for (int h = 0; h < H; ++h) {
    const int32_t f = g->tied_f[h] >= 0 ? g->tied_f[h] * (int32_t)rb : -1, sl = g->tied_s[h] >= 0 ? g->tied_s[h] * (int32_t)rb : -1;
    mi[h] = SlabRow{h, 0, f, sl, g->tied_w[h], g->work_pi[h], K[h], 0.f};
    mo[h] = SlabRow{h, 0, f, sl, g->tied_w[h], g->work_pi[h], 0.f, 0.f};
    g->big_f_off[h] = f;
  }
  make_slab_list(2, G, in, mi, bundles_by_length(in, G), &g->big_in);
  make_slab_list(2, G, outl, mo, bundles_by_length(outl, G), &g->big_out);
  g->big_pdf = SlabListHost();
  float sum_pi = 0.f;
  for (int h = 0; h < H; ++h) sum_pi += g->work_pi[h];
  g->big_sum_pi = sum_pi;
}

int build_schedules(tc_den_graph *g) {
  bool want_big = debug_flag(kDbgForceStreamed) || g->H > kMaxSplitPositions || g->P > kMaxIndex;
  bool split_made = false;  // tied only thanks to make_work_graph
  // ---- the tied path: on the FST as it is, or on its tied-ified work graph
  std::vector<char> special;
  {
    g->work_H = g->H;
    g->work_src = g->arc_src;
    g->work_dst = g->arc_dst;
    g->work_pdf = g->arc_pdf;
    g->work_prob = g->arc_prob;
    g->work_pi = g->initial_probs;
    g->copy_first.resize(g->H + 1);
    std::iota(g->copy_first.begin(), g->copy_first.end(), 0);
    bool tied = !debug_flag(kDbgForceGeneral) && detect_tied(g, &special);
    if (!tied && !debug_flag(kDbgForceGeneral) && !debug_flag(kDbgNoSplit) && make_work_graph(g)) {
      tied = detect_tied(g, &special);
      split_made = tied;
    }
    g->tied = tied;
    // a general graph of more than kMaxIndex states fits no on-chip layout: straight to the streamed tables (the general
    // schedule builders pack state indices into 16 + 2 bits and would run for seconds before compute_layout refused)
    if (!tied && g->H > kMaxIndex) want_big = true;
  }
  auto restore_unsplit = [&]() {
    g->work_H = g->H;
    g->work_src = g->arc_src;
    g->work_dst = g->arc_dst;
    g->work_pdf = g->arc_pdf;
    g->work_prob = g->arc_prob;
    g->work_pi = g->initial_probs;
    g->copy_first.resize(g->H + 1);
    std::iota(g->copy_first.begin(), g->copy_first.end(), 0);
    g->tied = false;
    g->gen_owner = false;
    g->tied_fs.clear();
    g->tied_w.clear();
    g->fwd = ScheduleHost();
    g->bwd = ScheduleHost();
  };
  // General graphs of at most 8192 states (round 5): the owner-computes schedules of the tied kernels with the arc's pdf in
  // the cell (den_general_owner.hip) -- no in-band row cells, no barrier between walk and per-state pass, resident chunks.
  // What it does not take (rows of other lanes in the backward walk: out-degrees beyond 256; more than 8192 states; a
  // layout without alpha' in LDS) stays with round 1's kernel (den_kernels.hip).
  auto general_owner = [&]() {
    if (debug_flag(kDbgOldGeneral)) return false;
    const std::vector<char> none(g->work_src.size(), 0);
    int fit_row = 0;
    int64_t best_cells = -1;
    for (int max_row = kMaxRowLen; max_row <= 8 * kMaxRowLen; max_row *= 2) {
      if (!build_owner(g, none, max_row, true, true)) continue;
      const int64_t cells = g->fwd.padded_arcs + g->bwd.padded_arcs;
      if (best_cells < 0 || cells < best_cells) {
        best_cells = cells;
        fit_row = max_row;
      }
    }
    if (!fit_row || !build_owner(g, none, fit_row, false, true)) {
      g->fwd = ScheduleHost();
      g->bwd = ScheduleHost();
      return false;
    }
    g->gen_owner = true;
    g->layout_ok = true;
    return true;
  };
  // (A graph that is chain-structured only after state splitting could also run UNSPLIT on the general kernel above.  Measured at
  // 256 x 150, profiles/r05_split_vs_general.txt: 3000 states with 60 % of them entered through 2-3 pdfs 1.76 (split) vs 1.46 ms,
  // 6000 states 2.09 vs 1.88, 8000 states with 30 %: 2.08 vs 2.19, the phone-LM graph R3 3.45 vs 4.50 -- popular pdfs serialise the
  // general kernel's atomics.  Never more than 17 % either way at 256 sequences, while batches of at most 128 run the tied kernels
  // on two CUs per sequence, which the general kernel has no form of: the split stays the rule.)
  if (!want_big && g->tied) {
    // Rows longer than kMaxRowLen spill into secondary rows, each with a private accumulator slot in LDS; a graph
    // with many popular states (real phone-LM graphs: in-degrees of a hundred and more) and close to the 16384-state
    // limit may have no room for them, so the home rows are allowed to grow before the graph is given up to the
    // streamed kernels (8x slower per arc).
    // Which cut?  Of those that fit (counting passes: no placement work) the one that leaves the fewest cells: every
    // secondary row costs its wave a 64-lane slot that is mostly padding when the wave has few of them (R4: 7 % of the forward
    // cells with rows cut at 32, R3: 5 %), a long home row pads the 63 rows that run in lockstep with it.
    int fit_row = 0;
    {
      int64_t best_cells = -1;
      for (int max_row = kMaxRowLen; max_row <= 8 * kMaxRowLen; max_row *= 2) {
        if (!build_owner(g, special, max_row, true)) continue;
        const int64_t cells = g->fwd.padded_arcs + g->bwd.padded_arcs;
        if (best_cells < 0 || cells < best_cells) {
          best_cells = cells;
          fit_row = max_row;
        }
        if (debug_flag(kDbgSchedTrace)) fprintf(stderr, "[sched] rows cut at %d: %lld cells, %d bytes of LDS\n", max_row, (long long)cells, 4 * g->layout.total_floats);
      }
      if (fit_row && !build_owner(g, special, fit_row)) {
        fit_row = 0;  // (a full build can still fail where the count fitted: more than 256 chunks per wave)
        for (int max_row = kMaxRowLen; max_row <= 8 * kMaxRowLen && !fit_row; max_row *= 2)
          if (build_owner(g, special, max_row)) fit_row = max_row;
      }
    }
    if (fit_row) {
      g->layout_ok = true;
      return TC_OK;
    }
    if (split_made && g->H <= kMaxIndex) {
      // The graph became tied only through state splitting, and the enlarged work graph does not fit the owner-computes
      // layouts.  Before it is given up to the streamed kernels (~8x slower per arc), the ORIGINAL graph gets its
      // chance on the general on-chip kernel, which it may well fit.
      std::vector<char> special_split = special;
      std::vector<int32_t> ws = g->work_src, wd = g->work_dst, wp = g->work_pdf, cf = g->copy_first, tf = g->tied_f, ts = g->tied_s;
      std::vector<float> wpr = g->work_prob, wpi = g->work_pi, tw = g->tied_w;
      std::vector<uint32_t> tfs = g->tied_fs;
      const int32_t wH = g->work_H;
      restore_unsplit();
      if (general_owner()) return TC_OK;
      build_general(g);
      g->layout_ok = compute_layout(g->H, g->P, 256, std::max(g->fwd.extra_slots, g->bwd.extra_slots), false, &g->layout);
      if (g->layout_ok) return TC_OK;
      // no: back to the split graph for the streamed tied kernels
      g->work_H = wH;
      g->work_src.swap(ws);
      g->work_dst.swap(wd);
      g->work_pdf.swap(wp);
      g->work_prob.swap(wpr);
      g->work_pi.swap(wpi);
      g->copy_first.swap(cf);
      g->tied_f.swap(tf);
      g->tied_s.swap(ts);
      g->tied_w.swap(tw);
      g->tied_fs.swap(tfs);
      g->tied = true;
      special.swap(special_split);
    }
    want_big = true;  // tied but beyond the on-chip layouts: the streamed kernels keep the tied factorisation
  }
  if (!want_big) {
    if (g->work_H != g->H) restore_unsplit();  // (a split that did not make the graph chain-structured)
    if (general_owner()) return TC_OK;
    build_general(g);
    g->layout_ok = compute_layout(g->H, g->P, 256, std::max(g->fwd.extra_slots, g->bwd.extra_slots), false, &g->layout);
    if (g->layout_ok) return TC_OK;
  }
  // the per-frame working set does not fit LDS: stream it.  Slabs of 16 or 32 sequences: measured at 256 sequences
  // (profiles/r04_ablations.txt), 32 wins on large graphs although its slice no longer fits L2 -- fewer, longer waves
  // per frame and half the atomics (10-arc-per-state graphs of 28000 states: 20.7 vs 22.7 ms, 40000: 28.4 vs 30.5);
  // at 20000 states the two are equal, a 24000-state phone-LM-structured graph prefers 16 (19.1 vs 20.5 ms), and
  // below that the kernels are short and the narrow slabs' larger grids win (8192 states: 8.7 vs 10.2 ms).
  g->big = true;
  {
    const int64_t states = g->tied ? g->work_H : g->H;
    g->big_G = states >= 28000 ? 32 : 16;
    if (debug_flag(kDbgSlabWide)) g->big_G = 32;
    if (debug_flag(kDbgSlabNarrow)) g->big_G = 16;
  }
  g->layout_ok = false;
  g->fwd = ScheduleHost();
  g->bwd = ScheduleHost();
  if (g->tied)
    build_big_tied(g, special);
  else
    build_big(g);
  g->big_hb = (std::max(g->big_in.bundles, g->big_out.bundles) + g->big_G - 1) / g->big_G;
  return TC_OK;
}

}  // namespace tc

using namespace tc;

extern "C" {

int tc_debug_set(const char *key, int value) {
  static const char *const names[kDbgCount] = {"force_general", "force_streamed", "no_split", "no_pdf_banks",
                                               "no_bank_search", "sched_trace", "no_phase_split", "no_num_overlap", "no_pair",
                                               "force_pair", "no_tune", "no_mitm", "force_mitm", "slab_wide", "slab_narrow", "exp_per_frame", "old_arrange", "no_planes", "old_general", "phantom_pdf0", "no_pdf_search", "no_split_source", "planes_meet_at"};
  if (!key) return TC_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < kDbgCount; ++i)
    if (!strcmp(key, names[i])) {
      g_debug[i].store(value, std::memory_order_relaxed);
      return TC_OK;
    }
  return TC_ERR_INVALID_ARGUMENT;
}

// TORCHAIN_HIP_DEBUG="key[=value],key..." in the environment sets the same switches when the library is loaded (for
// runs that cannot call tc_debug_set, e.g. a profiler around an unmodified program).
namespace {
const bool g_env_applied = [] {
  const char *env = getenv("TORCHAIN_HIP_DEBUG");
  if (!env) return false;
  std::string all(env);
  size_t pos = 0;
  while (pos <= all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    std::string item = all.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    int value = 1;
    const size_t eq = item.find('=');
    if (eq != std::string::npos) {
      value = atoi(item.c_str() + eq + 1);
      item.resize(eq);
    }
    if (tc_debug_set(item.c_str(), value) != TC_OK) fprintf(stderr, "libtorchain_hip: unknown switch '%s' in TORCHAIN_HIP_DEBUG\n", item.c_str());
  }
  return true;
}();
}  // namespace

int64_t tc_debug_counter(const char *key) {
  if (key && !strcmp(key, "pool_device_allocs")) return pool_counter(0);
  if (key && !strcmp(key, "pool_reuses")) return pool_counter(1);
  static const char *const names[kCntCount] = {"den_launches", "den_backward_launches", "num_launches", "num_backward_launches", "layout_launches",
                                              "den_long_utterance_launches"};
  for (int i = 0; key && i < kCntCount; ++i)
    if (!strcmp(key, names[i])) return g_launches[i].load(std::memory_order_relaxed);
  return -1;
}

int tc_den_graph_create(tc_den_graph **out, int32_t num_states, int64_t num_arcs, const int32_t *arc_src,
                        const int32_t *arc_dst, const int32_t *arc_ilabel, const float *arc_weight,
                        const float *final_weight, int32_t start_state, int32_t num_pdfs) {
  if (!out) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (num_states <= 0 || num_arcs < 0 || num_pdfs <= 0 || !final_weight) return TC_ERR_INVALID_ARGUMENT;
  if (num_arcs > 0 && (!arc_src || !arc_dst || !arc_ilabel || !arc_weight)) return TC_ERR_INVALID_ARGUMENT;
  if (start_state < 0 || start_state >= num_states) return TC_ERR_BAD_FST;
  if (num_arcs >= (int64_t)1 << 30) return TC_ERR_UNSUPPORTED;
  for (int64_t a = 0; a < num_arcs; ++a) {
    if (arc_src[a] < 0 || arc_src[a] >= num_states || arc_dst[a] < 0 || arc_dst[a] >= num_states)
      return TC_ERR_BAD_FST;
    if (arc_ilabel[a] < 1 || arc_ilabel[a] > num_pdfs) return TC_ERR_BAD_FST;  // [K] KALDI_ASSERT on pdf_id
    if (a > 0 && arc_src[a] < arc_src[a - 1]) return TC_ERR_BAD_FST;           // must be state-major
    if (!(arc_weight[a] == arc_weight[a])) return TC_ERR_BAD_FST;
  }
  tc_den_graph *g = new tc_den_graph();
  g->H = num_states;
  g->P = num_pdfs;
  g->A = num_arcs;
  g->arc_src.assign(arc_src, arc_src + num_arcs);
  g->arc_dst.assign(arc_dst, arc_dst + num_arcs);
  g->arc_pdf.resize(num_arcs);
  g->arc_prob.resize(num_arcs);
  for (int64_t a = 0; a < num_arcs; ++a) {
    g->arc_pdf[a] = arc_ilabel[a] - 1;
    g->arc_prob[a] = (float)std::exp(-(double)arc_weight[a]);  // [K] SetTransitions
  }
  // [K] DenominatorGraph::SetInitialProbs: start state only, 100 rounds of per-state-normalised HMM
  // propagation, averaged; double precision, stored as float.
  {
    const int H = num_states;
    std::vector<double> norm(H), cur(H, 0.0), nxt(H, 0.0), avg(H, 0.0);
    for (int s = 0; s < H; ++s) norm[s] = std::exp(-(double)final_weight[s]);
    for (int64_t a = 0; a < num_arcs; ++a) norm[arc_src[a]] += std::exp(-(double)arc_weight[a]);
    // [K] KALDI_ASSERT(tot_prob > 0): a state with no arcs and an infinite final weight would give norm = inf
    // and NaN initial probabilities
    for (int s = 0; s < H; ++s)
      if (!(norm[s] > 0.0) || !std::isfinite(norm[s])) {
        delete g;
        return TC_ERR_BAD_FST;
      }
    for (int s = 0; s < H; ++s) norm[s] = 1.0 / norm[s];
    cur[start_state] = 1.0;
    for (int iter = 0; iter < 100; ++iter) {
      for (int s = 0; s < H; ++s) avg[s] += cur[s] * (1.0 / 100);
      for (int64_t a = 0; a < num_arcs; ++a) {
        int s = arc_src[a];
        nxt[arc_dst[a]] += cur[s] * norm[s] * std::exp(-(double)arc_weight[a]);
      }
      double sum = 0.0;
      for (int s = 0; s < H; ++s) sum += nxt[s];
      for (int s = 0; s < H; ++s) {
        cur[s] = nxt[s] * (1.0 / sum);
        nxt[s] = 0.0;
      }
    }
    g->initial_probs.resize(H);
    for (int s = 0; s < H; ++s) g->initial_probs[s] = (float)avg[s];
  }
  build_schedules(g);
  *out = g;
  return TC_OK;
}

}  // extern "C"

// --- OpenFst binary VectorFst<StdArc> reader ------------------------------------------------------
namespace {
struct Reader {
  FILE *f;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string str() {
    int32_t n = get<int32_t>();
    if (!ok || n < 0 || n > (1 << 20)) {
      ok = false;
      return std::string();
    }
    std::string s((size_t)n, '\0');
    if (n && fread(&s[0], 1, (size_t)n, f) != (size_t)n) ok = false;
    return s;
  }
  void skip_symbol_table() {
    int32_t magic = get<int32_t>();
    if (magic != 2125658996) {
      ok = false;
      return;
    }
    str();                   // name
    get<int64_t>();          // available key
    int64_t size = get<int64_t>();
    for (int64_t i = 0; ok && i < size; ++i) {
      str();
      get<int64_t>();
    }
  }
};
}  // namespace

extern "C" {

int tc_den_graph_read(tc_den_graph **out, const char *rxfilename, int32_t num_pdfs) {
  if (!out || !rxfilename) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  // Kaldi rxfilename forms ([K] ParseInputPath): a plain path, or "command |" whose output is read
  // (e.g. "gunzip -c den.fst.gz |"); "-" / offsets into archives are not needed for a den.fst
  std::string name(rxfilename);
  while (!name.empty() && isspace((unsigned char)name.back())) name.pop_back();
  const bool is_pipe = !name.empty() && name.back() == '|';
  if (is_pipe) name.pop_back();
  FILE *f = is_pipe ? popen(name.c_str(), "r") : fopen(name.c_str(), "rb");
  if (!f) return TC_ERR_IO;
  Reader r{f};
  int rc = TC_OK;
  std::vector<int32_t> src, dst, il;
  std::vector<float> w, fin;
  int64_t start = 0, nstates = 0;
  try {
  do {
    if (r.get<int32_t>() != 2125659606) { rc = TC_ERR_IO; break; }  // kFstMagicNumber
    std::string fsttype = r.str(), arctype = r.str();
    int32_t version = r.get<int32_t>(), flags = r.get<int32_t>();
    r.get<uint64_t>();  // properties
    start = r.get<int64_t>();
    nstates = r.get<int64_t>();
    int64_t narcs = r.get<int64_t>();
    if (!r.ok || fsttype != "vector" || arctype != "standard" || version < 2 || (flags & 4) /* aligned */ ||
        nstates <= 0 || nstates > (1 << 24)) {
      rc = TC_ERR_IO;
      break;
    }
    if (flags & 1) r.skip_symbol_table();
    if (flags & 2) r.skip_symbol_table();
    // (the header's arc count is a hint, and in a damaged file anything: never allocated from unchecked)
    if (narcs > 0 && narcs <= (1 << 24)) { src.reserve(narcs); dst.reserve(narcs); il.reserve(narcs); w.reserve(narcs); }
    fin.resize(nstates);
    for (int64_t s = 0; r.ok && s < nstates; ++s) {
      fin[s] = r.get<float>();
      int64_t n = r.get<int64_t>();
      if (!r.ok || n < 0 || n > (1 << 28)) { r.ok = false; break; }
      for (int64_t i = 0; r.ok && i < n; ++i) {
        int32_t ilabel = r.get<int32_t>();
        r.get<int32_t>();  // olabel (== ilabel in a den.fst)
        float weight = r.get<float>();
        int32_t next = r.get<int32_t>();
        src.push_back((int32_t)s);
        dst.push_back(next);
        il.push_back(ilabel);
        w.push_back(weight);
      }
    }
    if (!r.ok) rc = TC_ERR_IO;
  } while (0);
  } catch (...) {  // (out of memory on a damaged file's sizes: no exception crosses the C boundary)
    rc = TC_ERR_IO;
  }
  if (is_pipe) {
    if (pclose(f) != 0 && rc == TC_OK) rc = TC_ERR_IO;
  } else {
    fclose(f);
  }
  if (rc != TC_OK) return rc;
  return tc_den_graph_create(out, (int32_t)nstates, (int64_t)src.size(), src.data(), dst.data(), il.data(), w.data(),
                             fin.data(), (int32_t)start, num_pdfs);
}

void tc_den_graph_free(tc_den_graph *g) {
  if (!g) return;
  for (auto &kv : g->dev) {
    if (kv.second.blob) {
      int cur = 0;
      if (hipGetDevice(&cur) == hipSuccess) {
        (void)hipSetDevice(kv.first);
        (void)hipFree(kv.second.blob);
        (void)hipSetDevice(cur);
      }
    }
  }
  delete g;
}

int32_t tc_den_graph_num_states(const tc_den_graph *g) { return g ? g->H : 0; }
int64_t tc_den_graph_num_arcs(const tc_den_graph *g) { return g ? g->A : 0; }
int32_t tc_den_graph_num_pdfs(const tc_den_graph *g) { return g ? g->P : 0; }

int tc_den_graph_initial_probs(const tc_den_graph *g, float *out_host) {
  if (!g || !out_host) return TC_ERR_INVALID_ARGUMENT;
  memcpy(out_host, g->initial_probs.data(), sizeof(float) * (size_t)g->H);
  return TC_OK;
}

int tc_den_graph_stats(const tc_den_graph *g, int64_t *o) {
  if (!g || !o) return TC_ERR_INVALID_ARGUMENT;
  o[0] = g->fwd.padded_arcs;
  o[1] = g->bwd.padded_arcs;
  o[2] = g->layout_ok ? layout_lds_bytes(g->layout, 150) : -1;
  o[3] = kThreads;
  o[4] = g->fwd.rows;
  o[5] = g->bwd.rows;
  o[6] = g->fwd.conflict_free_cost ? 1000 * g->fwd.conflict_cost / g->fwd.conflict_free_cost : 0;
  o[7] = g->bwd.conflict_free_cost ? 1000 * g->bwd.conflict_cost / g->bwd.conflict_free_cost : 0;
  o[8] = g->big ? 2 : (g->tied ? 1 : 0);
  return TC_OK;
}

// Host-side replay of the schedules (diagnostic; never on the product path): performs one arc walk the
// way the kernels consume the streams -- cell order, row-end masks, commit cursor, secondary rows and
// fix-up lists, position permutation, per-state tables -- and returns per state
//   direction 0: sum over in-arcs  (h -> g) of w * gather[h] * pdf_factor[pdf]
//   direction 1: sum over out-arcs (h -> g) of w * gather[g] * pdf_factor[pdf]
// so that tests without a GPU can compare the built schedules with the definition.
int tc_den_graph_debug_walk(const tc_den_graph *g, int direction, const float *gather, const float *pdf_factor,
                            float *out) {
  if (!g || !gather || !pdf_factor || !out || direction < 0 || direction > 1) return TC_ERR_INVALID_ARGUMENT;
  const int H = g->H;
  // streamed path: the lists are replayed the way the kernels read them (bundle, group of 16 lanes, chunk, step)
  auto slab_sums = [&](const SlabListHost &L, auto term, std::vector<float> *sums) {
    const int spc = 16 / L.W, NG = 64 / L.G, CH = NG * 16;
    for (int32_t b = 0; b < L.bundles; ++b) {
      const bool shared = L.head[(size_t)b * 2 + 1] < 0;  // one long row dealt to the groups: the kernels add the groups' sums
      float gsum[4] = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < NG; ++q) {
        const SlabRow &r = L.rows[(size_t)b * NG + q];
        if (r.row < 0 && !shared) continue;
        float sum = 0.f;
        for (int i = 0; i < r.n; ++i) sum += term(&L.rec[((size_t)L.head[(size_t)b * 2] + i / spc) * CH + q * 16 + (i % spc) * L.W]);
        if (shared)
          gsum[q] = sum;
        else
          (*sums)[r.row] = sum;
      }
      if (shared) (*sums)[L.rows[(size_t)b * NG].row] = NG == 4 ? (gsum[0] + gsum[1]) + (gsum[2] + gsum[3]) : gsum[0] + gsum[1];
    }
  };
  auto u2f = [](uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  if (g->big && g->tied) {
    // work-graph states; a split state's forward value sits on its first copy, its backward value on all
    const SlabListHost &L = direction == 0 ? g->big_in : g->big_out;
    auto pf = [&](int pdf) { return pdf >= 0 ? pdf_factor[pdf] : 0.f; };
    std::vector<float> val(g->work_H, 0.f), sums(g->work_H, 0.f);
    for (int h = 0; h < H; ++h)
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c)
        val[c] = direction == 0 ? (c == g->copy_first[h] ? gather[h] : 0.f) : gather[h] * pf(g->tied_f[c]);
    slab_sums(L, [&](const uint32_t *e) { return u2f(e[1]) * val[e[0] / (4 * L.G)]; }, &sums);
    for (int h = 0; h < H; ++h) {
      float total = 0.f;
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c) {
        const float own = direction == 0 ? (c == g->copy_first[h] ? gather[h] : 0.f) : gather[h];
        const float self = pf(g->tied_s[c]) * g->tied_w[c] * own;
        const float v = direction == 0 ? pf(g->tied_f[c]) * sums[c] + self : sums[c] + self;
        if (direction == 1) {
          total = v;
          break;
        }
        total += v;
      }
      out[h] = total;
    }
    return TC_OK;
  }
  if (g->big) {
    std::vector<float> sums(H, 0.f);
    slab_sums(direction == 0 ? g->big_in : g->big_out,
              [&](const uint32_t *e) { return u2f(e[2]) * gather[e[0] / (4 * g->big_G)] * pdf_factor[e[1] / (4 * g->big_G)]; }, &sums);
    for (int h = 0; h < H; ++h) out[h] = sums[h];
    return TC_OK;
  }
  if (!g->layout_ok) return TC_ERR_UNSUPPORTED;
  const ScheduleHost &sc = direction == 0 ? g->fwd : g->bwd;
  const int Hs = g->layout.Hs;
  // (plane-wise form: the replay keeps one accumulator per position, where the kernel reuses four rows per wave)
  std::vector<float> acc((size_t)std::max(g->layout.acc_floats, Hs + 4 + sc.extra_slots) + 64, 0.f);
  if (g->gen_owner) {
    // owner-computes schedules of a general graph: 8-byte cells {w, position * 4 | pdf * 4 << 16}, [chunk][4 blocks][lane]
    const int K = Hs / kThreads;
    std::vector<float> src_pos((size_t)Hs + 4, 0.f);
    for (int h = 0; h < H; ++h) src_pos[g->pos[h]] = gather[h];
    for (int w = 0; w < kWaves; ++w) {
      const int first = sc.wave_range[w].x, n = sc.wave_range[w].y;
      for (int l = 0; l < 64; ++l) {
        const int tid = 64 * w + l;
        int k = 0;
        float a = 0.f;
        auto slot = [&]() { return k < K ? 4 * (tid + kThreads * (k >> 2)) + (k & 3) : Hs + 4 + 64 * (sc.extra_first[w] + (k - K)) + l; };
        for (int i = 0; i < n; ++i) {
          const size_t c = (size_t)first + i, chunk = c / 8, q = c % 8;
          const uint32_t *base = &sc.cells6[chunk * 4 * 64 * 4];
          const uint32_t x = base[((q / 4) * 64 + l) * 4 + (q % 4)], idx = base[((2 + q / 4) * 64 + l) * 4 + (q % 4)];
          float wgt;
          memcpy(&wgt, &x, 4);
          a += wgt * src_pos[(idx & 0xffffu) >> 2] * pdf_factor[idx >> 18];
          const uint32_t m = sc.masks[(size_t)w * sc.mask_stride + (i / 2) / 8];
          const int bit = (i / 2) % 8;
          if ((m >> ((i & 1) ? bit : 8 + bit)) & 1u) {  // flag B: the pair's second cell ends a row, flag A: its first
            acc[slot()] = a;
            ++k;
            a = 0.f;
          }
        }
      }
    }
    for (size_t t = 0; t + 1 < sc.fix_begin.size(); ++t)
      for (int e = sc.fix_begin[t]; e < sc.fix_begin[t + 1]; ++e) acc[sc.fix[e].x] += acc[sc.fix[e].y];
    for (int h = 0; h < H; ++h) out[h] = acc[g->pos[h]];
    return TC_OK;
  }
  if (g->tied) {
    const int K = Hs / kThreads;
    const bool pw = g->layout.planewise;
    const int subs = pw ? sc.subs : 1, halves = pw ? sc.halves : 1, subs_all = subs * halves;
    // split gather source (ScheduleHost::halves == 2): the walk is replayed a half at a time, as the kernel makes it -- the
    // half's cells address positions inside the half, its secondary rows are folded from slots the next half reuses, and the
    // halves' row sums add up
    const int half_pos = halves == 2 ? 4096 * ((Hs / 4096 + 1) / 2) : 0;
    std::vector<float> total(acc.size(), 0.f);
    std::vector<float> src_pos((size_t)Hs + 4, 0.f);
    // a split state's alpha is the sum of its copies' (forward: the first copy carries the value), its
    // beta is shared by all copies (backward: every copy presents it)
    for (int h = 0; h < H; ++h)
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c) {
        const uint32_t fs = g->tied_fs[g->pos[c]];
        src_pos[g->pos[c]] = direction == 0 ? (c == g->copy_first[h] ? gather[h] : 0.f)
                                            : gather[h] * pdf_factor[(fs & 0xffffu) >> 2];
      }
    for (int hv = 0; hv < halves; ++hv) {
    std::fill(acc.begin(), acc.end(), 0.f);
    const int pos_base = hv ? half_pos : 0;
    for (int w = 0; w < kWaves; ++w)
     for (int sub = 0; sub < subs; ++sub) {
      // (plane-wise form: .y is the sub-stream's END counted from the wave's first cell, and the mask bytes are the wave's)
      const int first = sc.wave_range[(size_t)w * subs_all + hv * subs + sub].x;
      const int rel = pw ? first - sc.wave_range[(size_t)w * subs_all].x : 0;
      const int n = sc.wave_range[(size_t)w * subs_all + hv * subs + sub].y - rel;
      for (int l = 0; l < 64; ++l) {
        const int tid = 64 * w + l;
        int k = !pw ? 0 : sub == 0 ? K : 4 * (sub - 1);  // (sub-stream 0: the wave's secondary rows)
        float ax = 0.f, ay = 0.f;
        auto slot = [&]() {
          return k < K ? 4 * (tid + kThreads * (k >> 2)) + (k & 3) : Hs + 4 + 64 * (sc.extra_first[w] + (k - K)) + l;
        };
        auto cell = [&](int i, float *wgt, int *position) {
          const size_t c = (size_t)first + i, chunk = c / 8, q = c % 8;
          const uint32_t *base = &sc.cells6[chunk * 3 * 64 * 4];
          uint32_t x = base[((q / 4) * 64 + l) * 4 + (q % 4)];
          memcpy(wgt, &x, 4);
          const uint32_t o = base[(2 * 64 + l) * 4 + q / 2];
          *position = pos_base + (int)(((q & 1) ? o >> 16 : o & 0xffffu) >> (pw ? 0 : 2));
        };
        for (int i = 0; i < n; i += 2) {
          float w0, w1;
          int p0, p1;
          cell(i, &w0, &p0);
          cell(i + 1, &w1, &p1);
          uint32_t m = pw ? 0u : sc.masks[((size_t)w * subs + sub) * sc.mask_stride + (i / 2) / 8];
          int bit = (i / 2) % 8;
          if (pw) {  // one byte per chunk of the wave's stream; re-stated in the pair form the loop below tests
            const int cell = rel + i, chunk = cell / 8;
            const uint32_t byte = (sc.masks[(size_t)w * sc.mask_stride + chunk / 4] >> (8 * (chunk % 4))) & 0xffu;
            m = (((byte >> (cell % 8)) & 1u) << 8) | ((byte >> (cell % 8 + 1)) & 1u);
            bit = 0;
          }
          if ((m >> (8 + bit)) & 1u) {  // the row ends with the pair's first cell
            acc[slot()] = (ax + w0 * src_pos[p0]) + ay;
            ++k;
            ax = 0.f;
            ay = w1 * src_pos[p1];
          } else {
            ax += w0 * src_pos[p0];
            ay += w1 * src_pos[p1];
          }
          if ((m >> bit) & 1u) {
            acc[slot()] = ax + ay;
            ++k;
            ax = ay = 0.f;
          }
        }
      }
    }
    const size_t lists = (sc.fix_begin.size() - 1) / (size_t)halves;  // (fix_begin: [half][thread][plane] in the plane-wise form)
    for (size_t t = hv * lists; t < (hv + 1) * lists; ++t)
      for (int e = sc.fix_begin[t]; e < sc.fix_begin[t + 1]; ++e) {
        const size_t tl = t - hv * lists;
        if (pw && ((sc.fix[e].x >> 2) % kThreads != (int)(tl / (K / 4)) || sc.fix[e].x / (4 * kThreads) != (int)(tl % (K / 4))))
          return TC_ERR_UNSUPPORTED;  // (an entry must sit in the list of the thread and plane that own its state)
        acc[sc.fix[e].x] += acc[sc.fix[e].y];
      }
    for (int p = 0; p < Hs; ++p) total[p] += acc[p];
    }  // halves
    acc.swap(total);
    for (int h = 0; h < H; ++h) {
      float sum = 0.f;
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c) {
        const int p = g->pos[c];
        const uint32_t fs = g->tied_fs[p];
        const float self = pdf_factor[fs >> 18] * g->tied_w[p] * (direction == 0 && c != g->copy_first[h] ? 0.f : gather[h]);
        const float v = direction == 0 ? pdf_factor[(fs & 0xffffu) >> 2] * acc[p] + self : acc[p] + self;
        if (direction == 1) {  // every copy computes the state's out-sum: take the first
          sum = v;
          break;
        }
        sum += v;
      }
      out[h] = sum;
    }
    return TC_OK;
  }
  // general schedules: 8-byte cells [pair][lane][2] with in-band ROW cells
  for (int w = 0; w < kWaves; ++w) {
    const int first = sc.wave_range[w].x, n = sc.wave_range[w].y;
    for (int l = 0; l < 64; ++l) {
      int cur = Hs;  // dummy row
      float a = 0.f;
      for (int i = 0; i < n; ++i) {
        const size_t c = (size_t)first + i;
        const ArcRec &r = sc.cells[((c >> 1) * 64 + l) * 2 + (c & 1)];
        if (r.idx & kRowFlag) {
          acc[cur] = a;
          uint32_t x;
          memcpy(&x, &r.w, 4);
          cur = (int)(x & 0xffffu);
          a = 0.f;
        } else {
          a += r.w * gather[std::min<uint32_t>(r.idx >> 18, (uint32_t)H - 1)] * pdf_factor[(r.idx >> 2) & 0x3fffu];
        }
      }
    }
  }
  for (int t = 0; t < kThreads; ++t)
    for (int e = sc.fix_begin[t]; e < sc.fix_begin[t + 1]; ++e) acc[sc.fix[e].x] += acc[sc.fix[e].y];
  for (int h = 0; h < H; ++h) out[h] = acc[h];
  return TC_OK;
}

static int upload_den_graph(tc_den_graph *g, int device);

// Uploads the graph's tables to `device` once; the first call for an eligible graph also times the fused and the
// two-sequence kernel on that device and keeps the faster (tune_den_variant).
int tc_den_graph_prepare(tc_den_graph *g, int device) {
  if (!g) return TC_ERR_INVALID_ARGUMENT;
  const int rc = upload_den_graph(g, device);
  if (rc != TC_OK) return rc;
  return tune_den_variant(g, device);
}

static int upload_den_graph(tc_den_graph *g, int device) {
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->dev.count(device)) return TC_OK;
  int prev = 0;
  TC_HIP_CHECK(hipGetDevice(&prev));
  TC_HIP_CHECK(hipSetDevice(device));
  const int Hs = (g->H + 3) & ~3;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  struct Part { const void *src; size_t bytes; size_t off; };
  if (g->big) {
    const std::vector<float> &pi_src = g->tied ? g->work_pi : g->initial_probs;
    std::vector<float> pi_pad(pi_src.size() + 8, 0.f);
    std::copy(pi_src.begin(), pi_src.end(), pi_pad.begin());
    const std::vector<int32_t> none_i(1, -1);
    const std::vector<float> none_f(1, 0.f);
    const bool tb = g->tied;
    const SlabListHost *lists[3] = {&g->big_in, &g->big_out, &g->big_pdf};
    const std::vector<SlabRow> no_rows(4, SlabRow{-1, 0, -1, -1, 0.f, 0.f, 0.f, 0.f});
    const std::vector<int32_t> no_head(2, 0);
    const std::vector<uint32_t> no_rec(256, 0u);
    Part parts[11];
    for (int l = 0; l < 3; ++l) {
      const bool have = lists[l]->bundles > 0;
      parts[3 * l] = Part{have ? (const void *)lists[l]->rows.data() : no_rows.data(), (have ? lists[l]->rows.size() : 4) * sizeof(SlabRow), 0};
      parts[3 * l + 1] = Part{have ? (const void *)lists[l]->head.data() : no_head.data(), (have ? lists[l]->head.size() : 2) * 4, 0};
      parts[3 * l + 2] = Part{have ? (const void *)lists[l]->rec.data() : no_rec.data(), (have ? lists[l]->rec.size() : 256) * 4, 0};
    }
    parts[9] = Part{pi_pad.data(), pi_pad.size() * 4, 0};
    parts[10] = Part{g->big_f_off.data(), g->big_f_off.size() * 4, 0};
    size_t total = 0;
    for (auto &p : parts) {
      p.off = total;
      total += align(p.bytes);
    }
    char *blob = nullptr;
    hipError_t e = hipMalloc((void **)&blob, total);
    if (e == hipSuccess)
      for (auto &p : parts) {
        e = hipMemcpy(blob + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
      }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return TC_ERR_HIP;
    }
    DenGraphDev d;
    d.blob = blob;
    SlabListDev *dl[3] = {&d.big.in, &d.big.out, &d.big.pdf};
    for (int l = 0; l < 3; ++l) {
      dl[l]->rows = (const SlabRow *)(blob + parts[3 * l].off);
      dl[l]->head = (const int2 *)(blob + parts[3 * l + 1].off);
      dl[l]->rec = (const uint32_t *)(blob + parts[3 * l + 2].off);
      dl[l]->bundles = lists[l]->bundles;
    }
    d.pi = (const float *)(blob + parts[9].off);
    d.big.tied = tb ? 1 : 0;
    d.big.G = g->big_G;
    d.big.in_blocks = (g->big_in.bundles + g->big_G - 1) / g->big_G;  // a block = 64 rows' worth of bundles = G bundles
    d.big.out_blocks = (g->big_out.bundles + g->big_G - 1) / g->big_G;
    d.big.hb = g->big_hb;
    d.big.f_off = (const int32_t *)(blob + parts[10].off);
    g->dev[device] = d;
    return TC_OK;
  }
  std::vector<float> pi_pad(Hs + 4, 0.f);
  std::copy(g->initial_probs.begin(), g->initial_probs.end(), pi_pad.begin());
  if (g->tied || g->gen_owner) pi_pad = g->pi_pos;  // position order (build_owner)
  const std::vector<uint32_t> no_mask(1, 0u);
  const std::vector<int32_t> no_extra(kWaves, 0);
  // den_tied_pair.hip gathers from a [position][2 sequences] array: the same streams with offsets = position * 8,
  // which fit 16 bits up to 8192 positions
  std::vector<uint32_t> pair_cells[2];
  const bool want_pair = g->tied && g->layout.Hs <= 8192 && !g->fwd.cells6.empty();
  if (want_pair)
    for (int dir = 0; dir < 2; ++dir) {
      const std::vector<uint32_t> &src = dir == 0 ? g->fwd.cells6 : g->bwd.cells6;
      pair_cells[dir] = src;
      for (size_t chunk = 0; chunk + 1 <= src.size() / (3 * 64 * 4); ++chunk)
        for (size_t i = 0; i < 64 * 4; ++i) {
          uint32_t &x = pair_cells[dir][chunk * 3 * 64 * 4 + 2 * 64 * 4 + i];
          x = ((x & 0xffffu) << 1) | ((x >> 16) << 17);
        }
    }
  const std::vector<uint32_t> no_pair(4, 0u);
  Part parts[] = {
      {g->fwd.cells6.empty() ? (const void *)g->fwd.cells.data() : (const void *)g->fwd.cells6.data(),
       g->fwd.cells6.empty() ? g->fwd.cells.size() * sizeof(ArcRec) : g->fwd.cells6.size() * 4, 0},
      {g->fwd.wave_range.data(), g->fwd.wave_range.size() * sizeof(int2), 0},
      {g->bwd.cells6.empty() ? (const void *)g->bwd.cells.data() : (const void *)g->bwd.cells6.data(),
       g->bwd.cells6.empty() ? g->bwd.cells.size() * sizeof(ArcRec) : g->bwd.cells6.size() * 4, 0},
      {g->bwd.wave_range.data(), g->bwd.wave_range.size() * sizeof(int2), 0},
      {pi_pad.data(), pi_pad.size() * 4, 0},
      {g->fwd.fix_begin.data(), g->fwd.fix_begin.size() * 4, 0},
      {g->fwd.fix.data(), g->fwd.fix.size() * sizeof(int2), 0},
      {g->bwd.fix_begin.data(), g->bwd.fix_begin.size() * 4, 0},
      {g->bwd.fix.data(), g->bwd.fix.size() * sizeof(int2), 0},
      {g->tied_fs.data(), g->tied_fs.size() * 4, 0},
      {g->tied_w.data(), g->tied_w.size() * 4, 0},
      {g->fwd.masks.empty() ? no_mask.data() : g->fwd.masks.data(), std::max<size_t>(1, g->fwd.masks.size()) * 4, 0},
      {g->bwd.masks.empty() ? no_mask.data() : g->bwd.masks.data(), std::max<size_t>(1, g->bwd.masks.size()) * 4, 0},
      {g->fwd.extra_first.empty() ? no_extra.data() : g->fwd.extra_first.data(), kWaves * 4, 0},
      {g->bwd.extra_first.empty() ? no_extra.data() : g->bwd.extra_first.data(), kWaves * 4, 0},
      {want_pair ? pair_cells[0].data() : no_pair.data(), (want_pair ? pair_cells[0].size() : no_pair.size()) * 4, 0},
      {want_pair ? pair_cells[1].data() : no_pair.data(), (want_pair ? pair_cells[1].size() : no_pair.size()) * 4, 0},
  };
  size_t total = 0;
  for (auto &p : parts) {
    p.off = total;
    total += align(p.bytes);
  }
  char *blob = nullptr;
  hipError_t e = hipMalloc((void **)&blob, total);
  if (e == hipSuccess) {
    for (auto &p : parts) {
      e = hipMemcpy(blob + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) break;
    }
  }
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return TC_ERR_HIP;
  }
  DenGraphDev d;
  d.blob = blob;
  d.fwd = ScheduleDev{(const ArcRec *)(blob + parts[0].off), (const int2 *)(blob + parts[1].off),
                      (const int32_t *)(blob + parts[5].off), (const int2 *)(blob + parts[6].off),
                      (const uint32_t *)(blob + parts[11].off), (const int32_t *)(blob + parts[13].off),
                      g->fwd.mask_stride, g->fwd.nfix};
  d.bwd = ScheduleDev{(const ArcRec *)(blob + parts[2].off), (const int2 *)(blob + parts[3].off),
                      (const int32_t *)(blob + parts[7].off), (const int2 *)(blob + parts[8].off),
                      (const uint32_t *)(blob + parts[12].off), (const int32_t *)(blob + parts[14].off),
                      g->bwd.mask_stride, g->bwd.nfix};
  d.fwd.subs = g->fwd.subs;
  d.bwd.subs = g->bwd.subs;
  d.fwd.halves = g->fwd.halves;
  d.bwd.halves = g->bwd.halves;
  d.pi = (const float *)(blob + parts[4].off);
  if (g->tied) {
    d.tied_fs = (const uint32_t *)(blob + parts[9].off);
    d.tied_w = (const float *)(blob + parts[10].off);
  }
  if (want_pair) {
    d.fwd.cells_pair = blob + parts[15].off;
    d.bwd.cells_pair = blob + parts[16].off;
  }
  g->dev[device] = d;
  return TC_OK;
}

const char *tc_strerror(int code) {
  switch (code) {
    case TC_OK: return "ok";
    case TC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TC_ERR_BAD_FST: return "malformed FST";
    case TC_ERR_UNSUPPORTED: return "problem size not supported by the on-chip layout";
    case TC_ERR_WORKSPACE: return "workspace missing or too small";
    case TC_ERR_HIP: return "HIP runtime error";
    case TC_ERR_IO: return "cannot read OpenFst vector/standard file";
    case TC_ERR_NOT_SEPARABLE: return "merged supervision FST does not factor per sequence";
    default: return "unknown error";
  }
}

int tc_version(void) { return 100; }
int tc_last_hip_error(void) { return tc::g_last_hip_error; }

}  // extern "C"
