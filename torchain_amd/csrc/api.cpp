// C ABI entry points of the hot path (include/torchain_hip.h).  Replaces the reference's Kaldi
// bridge src/my_lib_chain.cpp:104-136: validates the borrowed tensors the way common::make_matrix
// does (2-D, unit column stride, src/common.hpp:109-117), carves the caller's workspace, and enqueues
// the kernels on the caller's stream.  No host synchronisation; no allocation once the per-device supervision
// pool (supervision.cpp) is warm.
#include <algorithm>
#include <string>
#include <cstring>

#include "chain_internal.h"

using namespace tc;

namespace tc {
int launch_den_mode(const DenParams &p, int accumulate, hipStream_t stream);
}

namespace {

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
  float *big_expy, *big_beta, *big_small, *big_y;  // streamed path only
  uint32_t *big_gam = nullptr;
  int big_exp_frames = 1;
  float *beta_hist, *fwd_norm, *bwd_norm;          // two-CU forms of small batches only (den_tied_split.hip, den_tied_mitm.hip)
  uint32_t *mitm_sync = nullptr;
  float *pair_norm = nullptr;                      // two-sequence form (den_tied_pair.hip)
  uint32_t *pair_sync = nullptr;
  long long *pair_stamps = nullptr;                // ... its diagnostic builds (-DTC_PAIR_STAMPS): raw cycle stamps
  float *alpha_hist, *asum_g;
  float *src_scratch = nullptr, *part_scratch = nullptr;  // plane-wise kernel with a split gather source (28673..40960 positions)
  double *den_lp, *num_lp, *y2, *xent_lp;
  float *ab, *gs;
  int32_t *fail;
  double *scalar;
  size_t total;
};

// Hs: stored states per frame of the alpha history (layout positions for tied graphs: build_owner)
int hist_states(const tc_den_graph *g) {
  if (g->big) return ((g->tied ? g->work_H : g->H) + 3) & ~3;
  return g->tied || g->gen_owner ? g->layout.Hs : ((g->H + 3) & ~3);
}
int big_p(const tc_den_graph *g) { return g->big ? g->P : 0; }
int big_h(const tc_den_graph *g) { return g->big ? (g->tied ? g->work_H : g->H) : 0; }  // tied: work-graph states
int big_g(const tc_den_graph *g) { return g->big ? g->big_G : 16; }
int big_hb(const tc_den_graph *g) { return g->big ? g->big_hb : 0; }
// Batches the tied on-chip kernel may run as two CUs per sequence get room for the second history.  (Whether
// they do is decided at launch: device size, layout for this T, diagnostic switch.)
bool split_room(const tc_den_graph *g, int S) { return g->tied && !g->big && S <= kSplitMaxSeq; }
// Tied on-chip graphs of at most 8192 positions may run two sequences per workgroup (den_tied_pair.hip): two more
// history rows, the two roles' normalisers and the pairing words.
// ... the plane-wise kernel of 16385..40960 positions (den_tied_planes.hip): scratch rows of its split-source form
bool planes_room(const tc_den_graph *g) { return g->tied && !g->big && g->layout_ok && g->layout.planewise; }
bool pair_room(const tc_den_graph *g) { return g->tied && !g->big && g->layout_ok && g->layout.JV == kJvSmall; }

// big_P != 0 selects the streamed path's layout: sequences padded to slabs of 16, [slab][state][16] matrices
Workspace carve(char *base, int Hs, int S, int T, int big_P = 0, int big_H = 0, int big_G = 16, int big_hb = 0, bool split = false, bool pair = false,
                bool planes = false) {
  Workspace w;
  const int Sp = (S + big_G - 1) / big_G * big_G;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  // (pair form: row T + 1 holds the backward role's B_M)
  w.alpha_hist = (float *)take(big_P ? (size_t)(T + 1) * Sp * big_H * sizeof(float) : (size_t)(T + (pair ? 2 : 1)) * S * Hs * sizeof(float));
  w.asum_g = (float *)take((size_t)S * asum_stride(T) * sizeof(float));  // frame sums of utterances too long for LDS
  if (planes && Hs > kMaxPlanePositions) {
    const int planes_n = Hs / 4096, planes_b = planes_n - (planes_n + 1) / 2;
    // (per WORKGROUP: two per sequence in the two-workgroup form of small batches)
    w.src_scratch = (float *)take((size_t)2 * S * 4096 * planes_b * sizeof(float));
    w.part_scratch = (float *)take((size_t)2 * S * (Hs + 4096 * ((planes_n + 1) / 2)) * sizeof(float));  // (+ alpha' of the first half's planes)
  }
  w.den_lp = (double *)take((size_t)S * 8);
  w.num_lp = (double *)take((size_t)S * 8);
  w.xent_lp = (double *)take((size_t)S * 8);
  w.y2 = (double *)take((size_t)S * 8);
  w.ab = (float *)take((size_t)S * 4);
  w.gs = (float *)take((size_t)S * 4);
  w.fail = (int32_t *)take(256);
  w.scalar = (double *)take(4096);  // also the stamp area of diagnostic builds
  // streamed path: exp(y) of every frame, transposed once and used by both passes, when that is at most 1 GB
  // (else one frame at a time, recomputed by the backward pass: 2-7 % slower, measured; the cap keeps the workspace
  // of exactly the largest graphs from growing by gigabytes -- include/torchain_hip.h states the sizes)
  // (the diagnostic switch exp_per_frame selects the one-frame form at LAUNCH; the workspace is sized for the larger
  // layout whatever the switch says, so a size a caller cached stays valid when the switch changes)
  const int exp_frames_room = big_P && (size_t)T * Sp * big_P * sizeof(float) <= ((size_t)1 << 30) ? T : 1;
  w.big_exp_frames = debug_flag(kDbgExpPerFrame) ? 1 : exp_frames_room;
  w.big_expy = big_P ? (float *)take((size_t)exp_frames_room * Sp * big_P * sizeof(float)) : nullptr;
  w.big_beta = big_P ? (float *)take((size_t)2 * Sp * big_H * sizeof(float)) : nullptr;
  w.big_small = big_P ? (float *)take((size_t)big_small_floats(big_hb, big_P, T, Sp) * sizeof(float)) : nullptr;
  w.big_y = big_P ? (float *)take((size_t)Sp * big_H * sizeof(float)) : nullptr;
  w.big_gam = big_P ? (uint32_t *)take((size_t)Sp * big_P * sizeof(uint32_t)) : nullptr;
  w.beta_hist = split ? (float *)take((size_t)(T + 1) * S * Hs * sizeof(float)) : nullptr;
  w.fwd_norm = split ? (float *)take((size_t)S * (T + 2) * sizeof(float)) : nullptr;
  w.bwd_norm = split ? (float *)take((size_t)S * (T + 1) * sizeof(float)) : nullptr;
  w.mitm_sync = split ? (uint32_t *)take(mitm_sync_bytes(S)) : nullptr;
  w.pair_sync = pair ? (uint32_t *)take(pair_sync_bytes(S)) : nullptr;
  w.pair_norm = pair ? (float *)take((size_t)2 * (S + 1) * pair_norm_stride(T) * sizeof(float)) : nullptr;  // (+ a spare row each)
  w.pair_stamps = pair ? (long long *)take(pair_stamp_bytes(T)) : nullptr;  // (the workspace's last block)
  w.total = off;
  return w;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) ok = false;
    if (ok && prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    if (prev == device) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int fill_den_params(tc_den_graph *g, int device, int32_t S, const float *y, int64_t rows, int32_t cols,
                    int64_t y_stride, float leaky, float deriv_weight, float l2_scale, float *deriv,
                    int64_t deriv_stride, const Workspace &w, DenParams *p, const DenGraphDev *tuning = nullptr) {
  if (!g || !y || S <= 0 || rows <= 0 || rows % S != 0) return TC_ERR_INVALID_ARGUMENT;
  if (cols != g->P || y_stride < cols) return TC_ERR_INVALID_ARGUMENT;
  if (deriv && deriv_stride < cols) return TC_ERR_INVALID_ARGUMENT;
  if (!(leaky > 0.0f && leaky < 1.0f)) return TC_ERR_INVALID_ARGUMENT;  // [K] KALDI_ASSERT in the ctor
  const int T = (int)(rows / S);
  DenGraphDev d;
  if (tuning) {
    d = *tuning;  // (tune_den_variant: the variant under test is in its pair_choice)
  } else {
    int rc = tc_den_graph_prepare(g, device);
    if (rc != TC_OK) return rc;
    std::lock_guard<std::mutex> lock(g->mu);
    d = g->dev[device];
  }
  bool tied = g->tied;
  // tied graphs address states by layout position (a multiple of 4096 of them, phantoms included)
  const int nstates = g->big ? (tied ? g->work_H : g->H) : (tied || g->gen_owner ? g->layout.Hs : g->H);
  if (g->big) {
    p->L = DenLayout();
    p->L.Hs = (nstates + 3) & ~3;
    p->L.Ps = (g->P + 3) & ~3;
    tied = false;  // the on-chip tied kernel's tables are not used; p->big.tied selects the streamed variant
  } else if (g->layout.planewise ? !compute_layout_planes(nstates, g->P, T, std::max(g->fwd.extra_slots, g->bwd.extra_slots), &p->L)
                                 : !compute_layout(nstates, g->P, T, std::max(g->fwd.extra_slots, g->bwd.extra_slots), tied, &p->L, g->gen_owner)) {
    return TC_ERR_UNSUPPORTED;  // decided at graph build with T = 256; a much longer T may not fit LDS
  }
  p->big = d.big;
  p->big_expy = w.big_expy;
  p->big_exp_stride = w.big_exp_frames > 1 ? (int64_t)((S + g->big_G - 1) / g->big_G * g->big_G) * g->P : 0;
  p->big_beta = w.big_beta;
  p->big_small = w.big_small;
  p->big_y = w.big_y;
  p->big_gam = w.big_gam;
  p->beta_hist = w.beta_hist;
  p->fwd_norm = w.fwd_norm;
  p->bwd_norm = w.bwd_norm;
  p->mitm_sync = w.mitm_sync;
  p->pair_norm = w.pair_norm;
  p->pair_sync = w.pair_sync;
  p->pair_stamps = w.pair_stamps;
  p->pair_extra_slots = std::max(g->fwd.extra_slots, g->bwd.extra_slots);
  p->pair_choice = d.pair_choice > 0 ? 1 : 0;
  p->big_Sp = (S + g->big_G - 1) / g->big_G * g->big_G;
  p->big_sum_pi = g->big_sum_pi;
  p->gen_owner = !g->big && g->gen_owner ? 1 : 0;
  p->tied_fs = tied ? d.tied_fs : nullptr;
  p->tied_w = tied ? d.tied_w : nullptr;
  p->fwd = d.fwd;
  p->bwd = d.bwd;
  p->pi = d.pi;
  p->y = y;
  p->y_stride = y_stride;
  p->deriv = deriv;
  p->deriv_stride = deriv_stride;
  p->alpha_hist = w.alpha_hist;
  p->asum_g = w.asum_g;
  p->src_scratch = w.src_scratch;
  p->part_scratch = w.part_scratch;
  p->seq_logprob = w.den_lp;
  p->seq_y2 = w.y2;
  p->seq_ab = w.ab;
  p->seq_gsum = w.gs;
  p->S = S;
  p->T = T;
  p->H = nstates;
  p->P = g->P;
  p->leaky = leaky;
  p->deriv_weight = deriv_weight;
  p->l2_scale = l2_scale;
  p->stamps = (long long *)w.scalar;
  p->y_vec = (cols % 4 == 0 && y_stride % 4 == 0 && aligned16(y)) ? 1 : 0;
  p->d_vec = (deriv && cols % 4 == 0 && deriv_stride % 4 == 0 && aligned16(deriv)) ? 1 : 0;
  return TC_OK;
}

int fill_num_params(tc_supervision *sup, int device, hipStream_t stream, const float *y, int64_t rows, int32_t cols,
                    int64_t y_stride, float *deriv, int64_t deriv_stride, float *xent, int64_t xent_stride,
                    double *seq_lp, NumParams *p) {
  if (!sup || !y) return TC_ERR_INVALID_ARGUMENT;
  if ((int64_t)sup->S * sup->T != rows || sup->P != cols || y_stride < cols) return TC_ERR_INVALID_ARGUMENT;
  if ((deriv && deriv_stride < cols) || (xent && xent_stride < cols)) return TC_ERR_INVALID_ARGUMENT;
  int rc = tc_supervision_prepare(sup, device, stream);
  if (rc != TC_OK) return rc;
  {
    std::lock_guard<std::mutex> lock(sup->mu);
    p->t = sup->dev[device];
  }
  p->y = y;
  p->y_stride = y_stride;
  p->deriv = deriv;
  p->deriv_stride = deriv_stride;
  p->xent = xent;
  p->xent_stride = xent_stride;
  p->seq_logprob = seq_lp;
  p->S = sup->S;
  p->T = sup->T;
  p->P = sup->P;
  p->weight = sup->weight;
  p->lds_states = (sup->tab.max_states + 1) & ~1;
  p->lds_arcs = (sup->tab.max_arcs + 3) & ~3;
  p->lds_uniq = (sup->tab.max_uniq + 3) & ~3;
  return TC_OK;
}


}  // namespace

namespace tc {

// Fused kernel or two-sequence kernel for this graph on this device?  Both are run on a zero-filled batch of one
// sequence per CU and kTuneFrames frames (their time does not depend on the values) in scratch memory that is freed
// again; the two-sequence kernel is kept when it is at least 3% faster.  Any failure here leaves the fused kernel.
constexpr int kTuneFrames = 48, kTuneWarmup = 60, kTuneRounds = 4;
constexpr double kPairMinArcsPerState = 11.0;

int tune_den_variant(tc_den_graph *g, int device) {
  DenGraphDev d;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    DenGraphDev &slot = g->dev[device];
    if (slot.pair_choice != -1) return TC_OK;
    slot.pair_choice = -2;  // (a concurrent caller runs the fused kernel meanwhile)
    d = slot;
  }
  int choice = 0;
  float ms[2] = {0.f, 0.f};
  int preset = -1;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    auto it = g->preset_variant.find(device);
    if (it != g->preset_variant.end()) preset = it->second;
  }
  auto finish = [&]() {
    std::lock_guard<std::mutex> lock(g->mu);
    DenGraphDev &slot = g->dev[device];
    slot.pair_choice = choice;
    slot.tune_ms[0] = ms[0];
    slot.tune_ms[1] = ms[1];
    return TC_OK;
  };
  if (!pair_room(g) || !d.fwd.cells_pair || debug_flag(kDbgNoPair) || debug_flag(kDbgNoTune) || debug_flag(kDbgForcePair))
    return finish();
  // The two-sequence kernel shares one walk between two sequences and pays for it in its gamma frames: measured, it never
  // wins below about 11-12 arcs per state (C3's 8: 8 % slower) -- such graphs keep the fused kernel without being timed
  // (no timing launches, no scratch).  Its frame bodies are its own statement of the tied frame (clamp, gamma scale and
  // fixed-point adds are den_tied_device.h's shared helpers); tests/test_gpu_tied.py holds it to the fused kernel's values.
  if ((double)g->work_src.size() < kPairMinArcsPerState * (double)std::max(1, g->work_H) && preset < 0) return finish();
  if (preset >= 0) {  // the caller's choice (a cache of an earlier run, or rank 0's): no timing launches
    choice = preset;
    return finish();
  }
  // the choice an earlier run measured for this graph on this kind of device: no timing launches, no scratch allocation
  std::string cache_key;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) cache_key = tuning_cache_key(tc_den_graph_hash(g), prop.name);
    int cached = 0;
    if (!cache_key.empty() && tuning_cache_get(cache_key, &cached)) {
      choice = cached;
      return finish();
    }
  }
  DeviceGuard guard(device);
  if (!guard.ok) return finish();
  int num_cus = 0;
  if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || num_cus < 2)
    return finish();
  const int S = num_cus & ~1, T = kTuneFrames, P = g->P;
  const Workspace w0 = carve(nullptr, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g));
  const size_t ybytes = (size_t)S * T * P * sizeof(float);
  char *mem = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipMalloc((void **)&mem, 2 * ybytes + w0.total + 512) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  ok = ok && hipMemsetAsync(mem, 0, 2 * ybytes + w0.total + 512, stream) == hipSuccess;
  if (ok) {
    float *y = (float *)mem, *deriv = (float *)(mem + ybytes);
    char *wsp = mem + 2 * ybytes;
    wsp += (256 - ((uintptr_t)wsp & 255)) & 255;
    const Workspace w = carve(wsp, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g));
    DenParams p[2];
    for (int variant = 0; variant < 2 && ok; ++variant) {
      d.pair_choice = variant;
      ok = fill_den_params(g, device, S, y, (int64_t)S * T, P, P, 0.1f, -1.0f, 0.f, deriv, P, w, &p[variant], &d) == TC_OK;
    }
    // Untimed launches until the clocks have settled (a graph usually arrives on an idle device), then the two
    // kernels in turn, the best of kRounds each.
    for (int i = 0; i < kTuneWarmup && ok; ++i) ok = launch_den_mode(p[i & 1], 0, stream) == TC_OK;
    ms[0] = ms[1] = 1e30f;
    for (int round = 0; round < kTuneRounds && ok; ++round)
      for (int variant = 0; variant < 2 && ok; ++variant) {
        float t = 0.f;
        ok = hipEventRecord(e0, stream) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) ok = launch_den_mode(p[variant], 0, stream) == TC_OK;
        ok = ok && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
        ok = ok && hipEventElapsedTime(&t, e0, e1) == hipSuccess;
        ms[variant] = std::min(ms[variant], 0.5f * t);
      }
    if (!ok) ms[0] = ms[1] = 0.f;
    if (ok && ms[1] < 0.97f * ms[0]) choice = 1;
  }
  if (stream) (void)hipStreamSynchronize(stream);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (stream) {
    side_streams_forget(stream);  // (launch_den_mode made an entry for the temporary stream)
    (void)hipStreamDestroy(stream);
  }
  if (mem) (void)hipFree(mem);
  (void)hipGetLastError();  // (a failed scratch allocation is not the caller's error)
  if (ms[0] > 0.f && ms[1] > 0.f && !cache_key.empty()) tuning_cache_put(cache_key, choice, ms[0], ms[1]);  // (really timed)
  return finish();
}

}  // namespace tc

extern "C" {

int tc_den_graph_tuning(tc_den_graph *g, int device, int32_t *two_sequence_kernel, float *fused_ms, float *two_sequence_ms) {
  if (!g) return TC_ERR_INVALID_ARGUMENT;
  const int rc = tc_den_graph_prepare(g, device);
  if (rc != TC_OK) return rc;
  std::lock_guard<std::mutex> lock(g->mu);
  const DenGraphDev &d = g->dev[device];
  if (two_sequence_kernel) *two_sequence_kernel = d.pair_choice > 0 ? 1 : 0;
  if (fused_ms) *fused_ms = d.tune_ms[0];
  if (two_sequence_ms) *two_sequence_ms = d.tune_ms[1];
  return TC_OK;
}

int tc_den_graph_set_variant(tc_den_graph *g, int device, int32_t two_sequence_kernel) {
  if (!g || two_sequence_kernel < -1 || two_sequence_kernel > 1) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(g->mu);
  if (two_sequence_kernel < 0)
    g->preset_variant.erase(device);
  else
    g->preset_variant[device] = two_sequence_kernel;
  auto it = g->dev.find(device);
  if (it != g->dev.end() && it->second.pair_choice != -2) {
    // already on the device: the choice applies from the next launch on (-1: timed again at the next prepare); a graph
    // the two-sequence kernel does not fit keeps the fused one whatever is asked
    const bool fits = pair_room(g) && it->second.fwd.cells_pair;
    it->second.pair_choice = two_sequence_kernel < 0 ? -1 : (fits ? two_sequence_kernel : 0);
  }
  return TC_OK;
}

uint64_t tc_den_graph_hash(const tc_den_graph *g) {
  if (!g) return 0;
  // FNV-1a over what the schedules are built from: sizes, arcs (source, destination, pdf, probability bits)
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  };
  mix(&g->H, sizeof(g->H));
  mix(&g->P, sizeof(g->P));
  mix(&g->A, sizeof(g->A));
  mix(g->arc_src.data(), g->arc_src.size() * 4);
  mix(g->arc_dst.data(), g->arc_dst.size() * 4);
  mix(g->arc_pdf.data(), g->arc_pdf.size() * 4);
  mix(g->arc_prob.data(), g->arc_prob.size() * 4);
  return h ? h : 1;
}

int64_t tc_chain_workspace_bytes(const tc_den_graph *g, int32_t S, int32_t T) {
  if (!g || S <= 0 || T <= 0) return TC_ERR_INVALID_ARGUMENT;
  return (int64_t)carve(nullptr, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g)).total;
}

int tc_den_forward_backward(tc_den_graph *g, int32_t S, const float *y, int64_t rows, int32_t cols,
                            int64_t y_stride, float leaky, float deriv_weight, float l2_scale, int accumulate,
                            float *deriv, int64_t deriv_stride, double *logprob_dev, int32_t *status_dev,
                            void *workspace, int64_t workspace_bytes, int device, void *stream_v) {
  if (!g || S <= 0 || rows <= 0 || rows % S != 0) return TC_ERR_INVALID_ARGUMENT;
  const int T = (int)(rows / S);
  Workspace w = carve((char *)workspace, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g));
  if (!workspace || (int64_t)w.total > workspace_bytes || !aligned16(workspace)) return TC_ERR_WORKSPACE;
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_v;
  DenParams p;
  int rc = fill_den_params(g, device, S, y, rows, cols, y_stride, leaky, deriv_weight, l2_scale, deriv, deriv_stride, w,
                           &p);
  if (rc != TC_OK) return rc;
  rc = launch_den_mode(p, accumulate, stream);
  if (rc != TC_OK) return rc;
  if (logprob_dev || status_dev)
    rc = launch_den_reduce(w.den_lp, deriv ? w.ab : nullptr, w.gs, S, logprob_dev, status_dev, stream);
  return rc;
}

int tc_num_forward_backward(tc_supervision *sup, const float *y, int64_t rows, int32_t cols, int64_t y_stride,
                            float *deriv, int64_t deriv_stride, double *logprob_dev, void *workspace,
                            int64_t workspace_bytes, int device, void *stream_v) {
  if (!sup) return TC_ERR_INVALID_ARGUMENT;
  const size_t need = align256((size_t)sup->S * 8);
  if (!workspace || (int64_t)need > workspace_bytes || !aligned16(workspace)) return TC_ERR_WORKSPACE;
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_v;
  NumParams p;
  int rc = fill_num_params(sup, device, stream, y, rows, cols, y_stride, deriv, deriv_stride, nullptr, 0,
                           (double *)workspace, &p);
  if (rc != TC_OK) return rc;
  rc = launch_num(p, stream);
  if (rc != TC_OK) return rc;
  if (logprob_dev) rc = launch_sum_double((const double *)workspace, sup->S, (double)sup->weight, logprob_dev, stream);
  if (rc != TC_OK) return rc;
  return supervision_mark_use(sup, device, stream);
}

}  // extern "C"

// deriv_scale = 1: the reference's outputs.  deriv_scale = -1: what its backward returns (tc_chain_objf_and_grad).
// bct_input (tc_chain_step, (B, C, T) tensors): `y` is scratch that this call fills from bct_input (tc_to2d) in front of the
// denominator; the numerator, which reads y at the supervision's (frame, pdf) pairs only, reads bct_input where it lies
// and runs BESIDE that copy and the denominator on the side stream (its posteriors wait in the supervision's staging
// area for the scatter behind the denominator, as for small batches).
// xent_out / xent_objf_dev (tc_chain_step): the cross-entropy output and where sum(xent_out * xent) goes -- formed from
// the numerator's posteriors as they are written, not from the dense matrices
static int chain_objf(tc_den_graph *g, tc_supervision *sup, const float *y, int64_t rows, int32_t cols,
                      int64_t y_stride, float *results_dev3, float *deriv, int64_t deriv_stride, float *xent,
                      int64_t xent_stride, float l2_regularize, float leaky, float deriv_scale, float xent_scale,
                      void *workspace, int64_t workspace_bytes, int device, void *stream_v,
                      const float *xent_out = nullptr, int64_t xent_out_stride = 0, double *xent_objf_dev = nullptr,
                      int xent_bct = 0, int xent_out_bct = 0, const float *bct_input = nullptr,
                      bool xent_sums_ready = false, float *loss_dev1 = nullptr) {
  if (!g || !sup || !y || !results_dev3) return TC_ERR_INVALID_ARGUMENT;
  if (sup->P != g->P) return TC_ERR_INVALID_ARGUMENT;
  if ((int64_t)sup->S * sup->T != rows) return TC_ERR_INVALID_ARGUMENT;
  Workspace w = carve((char *)workspace, hist_states(g), sup->S, sup->T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, sup->S), pair_room(g), planes_room(g));
  if (!workspace || (int64_t)w.total > workspace_bytes || !aligned16(workspace)) return TC_ERR_WORKSPACE;
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_v;

  const float wgt = sup->weight;
  DenParams dp;
  // (deriv_scale is +-1: the products below are exact, so the scaled outputs are the exact negatives)
  int rc = fill_den_params(g, device, sup->S, y, rows, cols, y_stride, leaky, deriv_scale * -wgt,
                           deriv_scale * (wgt * l2_regularize), deriv, deriv_stride, w, &dp);
  if (rc != TC_OK) return rc;
  NumParams np;
  rc = fill_num_params(sup, device, stream, y, rows, cols, y_stride, deriv, deriv_stride, xent, xent_stride, w.num_lp,
                       &np);
  if (rc != TC_OK) return rc;
  np.deriv_scale = deriv_scale;
  np.xent_scale = xent_scale;
  if (xent_out && xent_objf_dev) {  // (also without the dense xent matrix: an evaluation step's cross-entropy objective)
    np.xent_out = xent_out;
    np.xent_out_stride = xent_out_stride;
    np.xent_out_bct = xent_out_bct;
    np.seq_xent = w.xent_lp;
  }
  np.xent_bct = xent_bct;  // (a (B, C, T) tensor the caller has cleared: only the posteriors' entries are written)
  if (bct_input) {
    np.y = bct_input;
    np.y_bct = 1;
  }
  auto fill_y = [&]() -> int {  // the frame-major copy the denominator reads
    return bct_input ? launch_layout(true, bct_input, const_cast<float *>(y), sup->S, cols, sup->T, y_stride, 1.0f, stream) : TC_OK;
  };

  SideStreams *ss = nullptr;
  rc = side_streams(stream, &ss);
  if (rc != TC_OK) return rc;
  if (xent && !xent_bct) {
    // xent_deriv is zero outside the numerator's posteriors.  The denominator kernels that write every row of deriv
    // anyway write these zero rows with them (under their arc walks, where stores cost next to nothing); the others get
    // a memset in front: 629 MB at C3, 0.13 ms.
    if (deriv && den_zeroes_xent(dp, ss->num_cus)) {
      dp.xent_zero = xent;
      dp.xent_stride = xent_stride;
      dp.x_vec = (cols % 4 == 0 && xent_stride % 4 == 0 && aligned16(xent)) ? 1 : 0;
    } else if (xent_stride == cols) {
      TC_HIP_CHECK(hipMemsetAsync(xent, 0, (size_t)rows * cols * sizeof(float), stream));
    } else {
      TC_HIP_CHECK(hipMemset2DAsync(xent, (size_t)xent_stride * 4, 0, (size_t)cols * 4, (size_t)rows, stream));
    }
  }
  // denominator first: it writes every element of deriv (-w*gamma_den - w*l2*y); the numerator then
  // adds its sparse posteriors.  [K] runs the numerator first; the sum is the same.
  // When the denominator leaves CUs idle (small batches) the numerator's recursion runs beside it on a side
  // stream, leaving its posteriors in the supervision's staging area; the scatter follows the denominator.
  // (a forward-only call overlaps the two forward recursions the same way; its numerator writes nothing but its sums)
  if ((bct_input || den_cus_used(dp, ss->num_cus) + 16 <= ss->num_cus) && np.t.stage && !debug_flag(kDbgNoNumOverlap)) {
    std::lock_guard<std::recursive_mutex> lock(ss->enqueue);
    np.staged = 1;
    TC_HIP_CHECK(hipEventRecord(ss->num_fork, stream));
    TC_HIP_CHECK(hipStreamWaitEvent(ss->num_side, ss->num_fork, 0));
    rc = launch_num(np, ss->num_side);
    if (rc == TC_OK) rc = fill_y();
    if (rc == TC_OK) rc = launch_den(dp, stream);
    // Join the side stream also when something failed after the fork: the caller sees an error and may free or reuse
    // the workspace and the supervision, which the numerator kernel could still be reading or writing -- and the
    // supervision's pool slot must learn of its last reader before it can be handed out again.
    hipError_t e = hipEventRecord(ss->num_join, ss->num_side);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, ss->num_join, 0);
    if (rc != TC_OK || e != hipSuccess) {
      (void)supervision_mark_use(sup, device, stream);
      if (rc != TC_OK) return rc;
      TC_HIP_CHECK(e);
    }
    rc = launch_num_scatter(np, stream);
    if (rc != TC_OK) {
      (void)supervision_mark_use(sup, device, stream);
      return rc;
    }
  } else {
    rc = fill_y();
    if (rc == TC_OK) rc = launch_den(dp, stream);
    if (rc != TC_OK) return rc;
    rc = launch_num(np, stream);
    if (rc != TC_OK) return rc;
  }
  rc = supervision_mark_use(sup, device, stream);
  if (rc != TC_OK) return rc;
  rc = launch_finalize(w.den_lp, w.num_lp, w.y2, w.ab, w.gs, sup->S, sup->T, wgt, l2_regularize, deriv != nullptr,
                       results_dev3, w.fail, stream, (np.seq_xent || xent_sums_ready) && xent_objf_dev ? w.xent_lp : nullptr,
                       xent_objf_dev, loss_dev1);
  if (rc != TC_OK) return rc;
  // (a (B, C, T) xent tensor is rows * cols contiguous floats: cleared as such)
  return launch_zero_on_fail(w.fail, deriv, deriv_stride, xent, xent_bct ? (int64_t)cols : xent_stride, y, y_stride,
                             deriv_scale * (wgt * l2_regularize), rows, cols, stream);
}

extern "C" {

int tc_chain_objf_and_deriv(tc_den_graph *g, tc_supervision *sup, const float *y, int64_t rows, int32_t cols,
                            int64_t y_stride, float *results_dev3, float *deriv, int64_t deriv_stride, float *xent,
                            int64_t xent_stride, float l2_regularize, float leaky, float xent_regularize,
                            void *workspace, int64_t workspace_bytes, int device, void *stream_v) {
  (void)xent_regularize;  // as in the reference it only decides whether the caller passes xent (my_lib_chain.cpp:127)
  return chain_objf(g, sup, y, rows, cols, y_stride, results_dev3, deriv, deriv_stride, xent, xent_stride, l2_regularize,
                    leaky, 1.0f, 1.0f, workspace, workspace_bytes, device, stream_v);
}

int tc_chain_objf_and_grad(tc_den_graph *g, tc_supervision *sup, const float *y, int64_t rows, int32_t cols,
                           int64_t y_stride, float *results_dev3, float *grad, int64_t grad_stride, float *xent_grad,
                           int64_t xent_stride, float l2_regularize, float leaky, float xent_regularize,
                           void *workspace, int64_t workspace_bytes, int device, void *stream_v) {
  return chain_objf(g, sup, y, rows, cols, y_stride, results_dev3, grad, grad_stride, xent_grad, xent_stride,
                    l2_regularize, leaky, -1.0f, -xent_regularize, workspace, workspace_bytes, device, stream_v);
}

// ---- one call per training step --------------------------------------------------------------------------------
// Workspace of tc_chain_step: [tc_chain_workspace_bytes | scratch of the xent objective | for (B, C, T) tensors the
// frame-major copies of input and gradient (and of xent_input and its gradient)].
namespace {
struct StepWorkspace {
  char *chain, *trace;
  float *y2d, *g2d, *x2d;
  size_t chain_bytes, total;
};
StepWorkspace carve_step(char *base, const tc_den_graph *g, int S, int T, int P, bool three_d, bool xent) {
  StepWorkspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  w.chain_bytes = carve(nullptr, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g)).total;
  w.chain = take(w.chain_bytes);
  w.trace = take((size_t)trace_workspace_bytes());
  const size_t mat = (size_t)S * T * P * sizeof(float);
  w.y2d = three_d ? (float *)take(mat) : nullptr;
  w.g2d = three_d ? (float *)take(mat) : nullptr;
  w.x2d = three_d && xent ? (float *)take(mat) : nullptr;
  w.total = off;
  return w;
}
}  // namespace

int64_t tc_chain_step_workspace_bytes(const tc_den_graph *g, int32_t S, int32_t T, int three_d, int with_xent) {
  if (!g || S <= 0 || T <= 0) return TC_ERR_INVALID_ARGUMENT;
  return (int64_t)carve_step(nullptr, g, S, T, g->P, three_d != 0, with_xent != 0).total;
}

int tc_chain_step(tc_den_graph *g, tc_supervision *sup, const float *input, const float *xent_input, int three_d,
                  int64_t row_stride, float l2_regularize, float leaky, float xent_regularize, int kaldi_way,
                  float *grad, float *xent_grad, float *results_dev3, float *loss_dev1, double *xent_objf_dev,
                  void *workspace, int64_t workspace_bytes, int device, void *stream_v) {
  if (!g || !sup || !input || !results_dev3) return TC_ERR_INVALID_ARGUMENT;
  const bool use_xent = xent_input != nullptr && xent_regularize != 0.0f;
  // grad == NULL: an evaluation step ([K] ComputeChainObjfAndDeriv with nnet_output_deriv == NULL, as Kaldi's own
  // diagnostics call it): the two forward recursions, the results and the loss; no Backward(), so no alpha-beta
  // check, no gradient written, no copy back.  (The reference's validation loop pays for a whole training step under
  // torch.no_grad(): example/chime5/train.py:150-171 -> torchain/functions.py:74,82 allocate and fill mmi_grad anyway.)
  const bool eval_only = grad == nullptr;
  if (eval_only && xent_grad) return TC_ERR_INVALID_ARGUMENT;
  if (!eval_only && use_xent && !xent_grad) return TC_ERR_INVALID_ARGUMENT;
  const int S = sup->S, T = sup->T, P = sup->P;
  if (P != g->P) return TC_ERR_INVALID_ARGUMENT;
  const int64_t rows = (int64_t)S * T;
  if (!three_d && row_stride < P) return TC_ERR_INVALID_ARGUMENT;
  const StepWorkspace w = carve_step((char *)workspace, g, S, T, P, three_d != 0, use_xent);
  if (!workspace || (int64_t)w.total > workspace_bytes || !aligned16(workspace)) return TC_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_v;
  // frame-major views of the tensors: the caller's own for 2-D input, copies for (B, C, T) (torchain/functions.py:118-125)
  const float *y = input, *xe = xent_input;
  float *gr = grad, *xg = use_xent ? xent_grad : nullptr;
  int64_t stride = row_stride, gstride = P;  // (2-D gradients are written as contiguous matrices)
  int rc = TC_OK;
  // (B, C, T) with the regulariser: xent_deriv has entries only where the numerator has posteriors, so it is written
  // straight into the caller's cleared (B, C, T) gradient, and the cross-entropy objective reads xent_input where it
  // lies -- no frame-major copy of either.
  // The reference's way (kaldi_way == 0, functions.py:96-103) makes a second objective call on xent_input INTO THE SAME
  // results, gradient and xent gradient: everything the first call wrote is overwritten.  Only that second call is made
  // here; what remains of the first is this library's own extra, the cross-entropy objective, which is defined on the
  // first call's xent_deriv: the numerator alone on `input` (no denominator, nothing written but the sequences' sums).
  const bool second_only = use_xent && !kaldi_way;
  const int bct = three_d && use_xent ? 1 : 0;
  if (three_d) {
    y = second_only ? w.x2d : w.y2d;  // (filled inside chain_objf, beside the numerator: bct_input)
    gr = eval_only ? nullptr : w.g2d;
    xg = use_xent && !eval_only ? xent_grad : nullptr;
    stride = P;
    if (use_xent && !eval_only) {
      DeviceGuard guard(device);
      if (!guard.ok) return TC_ERR_HIP;
      TC_HIP_CHECK(hipMemsetAsync(xent_grad, 0, (size_t)rows * P * sizeof(float), stream));
    }
  } else if (second_only) {
    y = xent_input;
  }
  (void)xe;
  // 2-D: the matrices leave as the reference's backward returns them (-deriv, -xent_regularize * xent_deriv); 3-D: the
  // MMI gradient's sign rides on the way back through tc_from2d, so the 2-D scratch holds the plain derivative
  const float dscale = three_d ? 1.0f : -1.0f, xscale = -xent_regularize;  // (the (B, C, T) xent gradient: final values too)
  // Kaldi's cross-entropy objective sum(xent_output * xent_deriv) ([K] nnet-chain-training.cc; a TODO in the reference,
  // torchain/functions.py:88-89): xent_objf_dev receives it times `xscale`, summed by the numerator over the entries it
  // writes (tc_xent_objf is the dense statement of the same sum)
  const Workspace wi = carve((char *)w.chain, hist_states(g), S, T, big_p(g), big_h(g), big_g(g), big_hb(g), split_room(g, S), pair_room(g), planes_room(g));
  if (second_only && xent_objf_dev) {
    DeviceGuard guard(device);
    if (!guard.ok) return TC_ERR_HIP;
    NumParams np;
    rc = fill_num_params(sup, device, stream, input, rows, P, three_d ? (int64_t)P : row_stride, nullptr, 0, nullptr, 0, wi.num_lp, &np);
    if (rc != TC_OK) return rc;
    np.y_bct = three_d ? 1 : 0;
    np.xent_scale = xscale;
    np.xent_out = xent_input;
    np.xent_out_stride = row_stride;
    np.xent_out_bct = three_d ? 1 : 0;
    np.seq_xent = wi.xent_lp;
    rc = launch_num(np, stream);
    if (rc == TC_OK) rc = supervision_mark_use(sup, device, stream);
    if (rc != TC_OK) return rc;
  }
  rc = chain_objf(g, sup, y, rows, P, stride, results_dev3, gr, gstride, xg, gstride, l2_regularize, leaky, dscale, xscale,
                  w.chain, (int64_t)w.chain_bytes, device, stream_v, use_xent && !second_only ? xent_input : nullptr, row_stride,
                  use_xent ? xent_objf_dev : nullptr, bct, bct, three_d ? (second_only ? xent_input : input) : nullptr,
                  second_only && xent_objf_dev != nullptr, loss_dev1);
  if (rc != TC_OK) return rc;
  if (three_d && !eval_only) {
    rc = tc_from2d(gr, P, S, P, T, -1.0f, grad, device, stream_v);
    if (rc != TC_OK) return rc;
  }
  return rc;  // (the loss value -objf / weight was written with the results: finalize_kernel)
}

int tc_xent_objf(const float *xent_output, int64_t rows, int32_t cols, int64_t output_stride, const float *xent_deriv,
                 int64_t deriv_stride, double *objf_dev, void *workspace, int64_t workspace_bytes, int device,
                 void *stream_v) {
  if (!xent_output || !xent_deriv || !objf_dev || rows <= 0 || cols <= 0 || output_stride < cols || deriv_stride < cols)
    return TC_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < trace_workspace_bytes() || !aligned16(workspace)) return TC_ERR_WORKSPACE;
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  return launch_trace_mat_mat(xent_output, output_stride, xent_deriv, deriv_stride, rows, cols, (double *)workspace,
                              objf_dev, (hipStream_t)stream_v);
}

int tc_to2d(const float *in_bct, int32_t B, int32_t Cn, int32_t T, float *out2d, int64_t out_stride, int device,
            void *stream_v) {
  if (!in_bct || !out2d || B <= 0 || Cn <= 0 || T <= 0 || out_stride < Cn) return TC_ERR_INVALID_ARGUMENT;
  if (B > 65535 || (T + 239) / 240 > 65535) return TC_ERR_UNSUPPORTED;  // grid y / z limits
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  return launch_layout(true, in_bct, out2d, B, Cn, T, out_stride, 1.0f, (hipStream_t)stream_v);
}

int tc_from2d(const float *in2d, int64_t in_stride, int32_t B, int32_t Cn, int32_t T, float scale, float *out_bct,
              int device, void *stream_v) {
  if (!in2d || !out_bct || B <= 0 || Cn <= 0 || T <= 0 || in_stride < Cn) return TC_ERR_INVALID_ARGUMENT;
  if (B > 65535 || (T + 239) / 240 > 65535) return TC_ERR_UNSUPPORTED;
  DeviceGuard guard(device);
  if (!guard.ok) return TC_ERR_HIP;
  return launch_layout(false, in2d, out_bct, B, Cn, T, in_stride, scale, (hipStream_t)stream_v);
}

}  // extern "C"
