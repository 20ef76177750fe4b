// Host side of the numerator supervision.  The reference hands Kaldi a heap copy of
// egs.outputs[0].supervision (src/my_lib_example.cpp:71-76) and Kaldi's NumeratorComputation walks
// the merged FST serially on the CPU.  Here the merged acceptor is split once, on the host, into its
// num_sequences independent per-sequence acceptors so that the HIP kernel can run one wavefront per
// sequence; the split is exact (see split comment below).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>

#include "chain_internal.h"

using namespace tc;

namespace {

// [K] ComputeFstStateTimes: returns the path length, or -1 if the FST is not an epsilon-free,
// time-sorted acceptor whose paths all have equal length.
int32_t state_times(int32_t n, const int32_t *arc_begin, const int32_t *ilabel, const int32_t *next,
                    const float *final_w, std::vector<int32_t> *times) {
  times->assign(n, -1);
  (*times)[0] = 0;
  int32_t total = -1;
  for (int32_t s = 0; s < n; ++s) {
    int32_t nt = (*times)[s] + 1;
    if (nt <= 0) return -1;
    for (int32_t a = arc_begin[s]; a < arc_begin[s + 1]; ++a) {
      if (ilabel[a] == 0) return -1;
      int32_t &ref = (*times)[next[a]];
      if (ref == -1)
        ref = nt;
      else if (ref != nt)
        return -1;
    }
    if (!std::isinf(final_w[s])) {
      if (total == -1)
        total = nt - 1;
      else if (total != nt - 1)
        return -1;
    }
  }
  return total;
}

}  // namespace

namespace tc {

// ---- per-device pool of supervision slots -------------------------------------------------------------
namespace {
struct DevPool {
  std::mutex mu;
  std::vector<PoolSlot *> idle;
};
// (process lifetime, never destroyed: a reader's look-ahead threads may stage supervisions while static destructors run at
// exit -- csrc/rand_reader.cpp, and the same reasoning as the example pool of csrc/egs_reader.cpp)
struct Pools {
  std::mutex mu;
  std::map<int, DevPool> by_device;
};
std::atomic<int64_t> g_pool_device_allocs{0}, g_pool_reuses{0};

DevPool &pool_of(int device) {
  static Pools *const pools = new Pools();
  std::lock_guard<std::mutex> lock(pools->mu);
  return pools->by_device[device];
}
}  // namespace

int64_t pool_counter(int which) { return which == 0 ? g_pool_device_allocs.load() : g_pool_reuses.load(); }

// Smallest idle slot that is large enough AND whose last recorded use has completed (hipEventQuery: no
// waiting); a new slot otherwise -- which happens only while the pool warms up.
int pool_acquire(int device, size_t bytes, PoolSlot **out) {
  DevPool &dp = pool_of(device);
  {
    std::lock_guard<std::mutex> lock(dp.mu);
    int best = -1;
    for (int i = 0; i < (int)dp.idle.size(); ++i) {
      PoolSlot *s = dp.idle[i];
      if (s->cap < bytes || (best >= 0 && s->cap >= dp.idle[best]->cap)) continue;
      if (hipEventQuery(s->done) != hipSuccess) continue;
      best = i;
    }
    if (best >= 0) {
      *out = dp.idle[best];
      dp.idle.erase(dp.idle.begin() + best);
      g_pool_reuses++;
      return TC_OK;
    }
  }
  int prev = 0;
  TC_HIP_CHECK(hipGetDevice(&prev));
  TC_HIP_CHECK(hipSetDevice(device));
  PoolSlot *s = new PoolSlot();
  s->cap = std::max<size_t>((bytes + bytes / 4 + 65535) & ~(size_t)65535, 1 << 20);  // room for the next, larger batch
  hipError_t e = hipMalloc((void **)&s->blob, s->cap);
  if (e == hipSuccess) e = hipHostMalloc((void **)&s->host, s->cap, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    if (s->blob) (void)hipFree(s->blob);
    if (s->host) (void)hipHostFree(s->host);
    delete s;
    return TC_ERR_HIP;
  }
  g_pool_device_allocs++;
  *out = s;
  return TC_OK;
}

void pool_release(int device, PoolSlot *slot) {
  if (!slot) return;
  DevPool &dp = pool_of(device);
  std::lock_guard<std::mutex> lock(dp.mu);
  dp.idle.push_back(slot);
}

// Every launch that reads a supervision's device tables moves the slot's `done` event behind itself on its
// stream, so the pool never hands the slot out while a kernel may still be reading it.
int supervision_mark_use(tc_supervision *sup, int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(sup->mu);
  auto it = sup->dev.find(device);
  if (it == sup->dev.end()) return TC_OK;
  // (not inside a graph capture: the event would become a captured one, useless to the pool's queries.  Whoever keeps the
  // graph keeps the supervision alive for as long as it may be replayed -- include/torchain_hip.h, "stream capture")
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive) return TC_OK;
  TC_HIP_CHECK(hipEventRecord(it->second.slot->done, stream));
  return TC_OK;
}

}  // namespace tc

extern "C" {

int tc_supervision_create(tc_supervision **out, float weight, int32_t S, int32_t T, int32_t label_dim,
                          int32_t num_states, const int32_t *arc_begin, const int32_t *arc_ilabel,
                          const float *arc_weight, const int32_t *arc_next, const float *final_weight) {
  if (!out) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (S <= 0 || T <= 0 || label_dim <= 0 || num_states <= 0 || !arc_begin || !final_weight)
    return TC_ERR_INVALID_ARGUMENT;
  const int32_t num_arcs = arc_begin[num_states];
  if (arc_begin[0] != 0 || num_arcs < 0) return TC_ERR_BAD_FST;
  if (num_arcs > 0 && (!arc_ilabel || !arc_weight || !arc_next)) return TC_ERR_INVALID_ARGUMENT;
  for (int32_t s = 0; s < num_states; ++s)
    if (arc_begin[s + 1] < arc_begin[s]) return TC_ERR_BAD_FST;
  for (int32_t a = 0; a < num_arcs; ++a) {
    if (arc_next[a] < 0 || arc_next[a] >= num_states) return TC_ERR_BAD_FST;
    if (arc_ilabel[a] < 1 || arc_ilabel[a] > label_dim) return TC_ERR_BAD_FST;  // [K] pdf_id range assert
  }
  std::vector<int32_t> times;
  if (state_times(num_states, arc_begin, arc_ilabel, arc_next, final_weight, &times) != S * T) return TC_ERR_BAD_FST;
  for (int32_t s = 1; s < num_states; ++s)
    if (times[s] < times[s - 1]) return TC_ERR_BAD_FST;  // states must be numbered in time order

  // first state of every time tau
  std::vector<int32_t> time_begin((size_t)S * T + 2, num_states);
  for (int32_t s = num_states - 1; s >= 0; --s) time_begin[times[s]] = s;
  time_begin[(size_t)S * T + 1] = num_states;
  for (int64_t tau = (int64_t)S * T; tau >= 0; --tau)
    if (time_begin[tau] > time_begin[tau + 1]) time_begin[tau] = time_begin[tau + 1];

  // Exact split at the sequence boundaries.  [K] AppendSupervision merges with fst::Concat +
  // RmEpsilon, so every state b_i at a boundary time q*T carries a copy of sequence q's start arcs
  // with a constant extra weight f_i (the final weight b_i had in sequence q-1).  Then
  //   Z_merged = prod_q Z_q  with  sequence q   : virtual start state with b_0's arcs,
  //                                sequence q-1 : b_i final with weight f_i - f_0,
  // and arc posteriors are unchanged.  Anything else is refused (TC_ERR_NOT_SEPARABLE).
  tc_supervision *sup = new tc_supervision();
  sup->weight = weight;
  sup->S = S;
  sup->T = T;
  sup->P = label_dim;
  NumTables &tb = sup->tab;
  tb.seq_state_off.push_back(0);
  tb.seq_arc_off.push_back(0);
  tb.seq_uniq_off.push_back(0);
  const float kInf = std::numeric_limits<float>::infinity();
  std::vector<float> boundary_final;  // final weights (tropical) of the states at time (q+1)*T for sequence q
  std::vector<int32_t> map_this_frame, touched, ucount;  // scratch of the unique-(frame, pdf) pass
  tb.arc_src.reserve(num_arcs);
  tb.arc_dst.reserve(num_arcs);
  tb.arc_logw.reserve(num_arcs);
  tb.arc_uniq.reserve(num_arcs);
  tb.in_arc.reserve(num_arcs);
  tb.uniq_arc.reserve(num_arcs);
  tb.final_logw.reserve((size_t)num_states + S);
  tb.out_begin.reserve((size_t)num_states + 2 * S);
  tb.in_begin.reserve((size_t)num_states + 2 * S);
  tb.level_begin.reserve((size_t)S * (T + 2));

  for (int32_t q = 0; q < S; ++q) {
    const int32_t b0 = time_begin[(size_t)q * T];           // first state at this sequence's start time
    const int32_t b1 = time_begin[(size_t)q * T + 1];       // one past the start-time states
    const int32_t e0 = time_begin[(size_t)(q + 1) * T];     // first state at the end time
    const int32_t e1 = time_begin[(size_t)(q + 1) * T + 1];
    if (q == 0 && b1 - b0 != 1) { delete sup; return TC_ERR_BAD_FST; }
    // final weights of the end-time states
    boundary_final.assign(e1 - e0, kInf);
    if (q == S - 1) {
      for (int32_t s = e0; s < e1; ++s) boundary_final[s - e0] = final_weight[s];
    } else {
      const int32_t ref = e0, nref = arc_begin[ref + 1] - arc_begin[ref];
      std::vector<std::pair<std::pair<int32_t, int32_t>, float>> ra, rb;
      for (int32_t a = arc_begin[ref]; a < arc_begin[ref + 1]; ++a)
        ra.push_back({{arc_ilabel[a], arc_next[a]}, arc_weight[a]});
      std::sort(ra.begin(), ra.end());
      for (int32_t s = e0; s < e1; ++s) {
        if (arc_begin[s + 1] - arc_begin[s] != nref || nref == 0) { delete sup; return TC_ERR_NOT_SEPARABLE; }
        rb.clear();
        for (int32_t a = arc_begin[s]; a < arc_begin[s + 1]; ++a)
          rb.push_back({{arc_ilabel[a], arc_next[a]}, arc_weight[a]});
        std::sort(rb.begin(), rb.end());
        const float f = rb[0].second - ra[0].second;
        for (int32_t i = 0; i < nref; ++i) {
          if (rb[i].first != ra[i].first) { delete sup; return TC_ERR_NOT_SEPARABLE; }
          float d = (rb[i].second - ra[i].second) - f;
          if (!(std::fabs(d) <= 1e-3f * (1.0f + std::fabs(f)))) { delete sup; return TC_ERR_NOT_SEPARABLE; }
        }
        boundary_final[s - e0] = f;
      }
    }
    // local numbering: local 0 = start; global state g in (b1 .. e1) -> g - b1 + 1
    const int32_t nloc = 1 + (e1 - b1);
    const int32_t sbase = (int32_t)tb.final_logw.size();
    const int32_t abase = (int32_t)tb.arc_src.size();
    auto local = [&](int32_t g) { return g - b1 + 1; };
    for (int32_t i = 0; i < nloc; ++i) tb.final_logw.push_back(-kInf);
    for (int32_t s = e0; s < e1; ++s) tb.final_logw[sbase + local(s)] = -boundary_final[s - e0];
    // level boundaries in local state ids (T + 2 entries)
    tb.level_begin.push_back(0);
    for (int32_t t = 1; t <= T; ++t) tb.level_begin.push_back(local(time_begin[(size_t)q * T + t]));
    tb.level_begin.push_back(nloc);
    // arcs, sorted by local source state; out_begin has nloc + 1 entries per sequence
    for (int32_t ls = 0; ls < nloc; ++ls) {
      const int32_t g = (ls == 0) ? b0 : (b1 + ls - 1);
      tb.out_begin.push_back((int32_t)tb.arc_src.size() - abase);
      if (ls > 0 && g >= e0) continue;  // end-time states: their arcs belong to the next sequence
      for (int32_t a = arc_begin[g]; a < arc_begin[g + 1]; ++a) {
        tb.arc_src.push_back(ls);
        tb.arc_dst.push_back(local(arc_next[a]));
        tb.arc_logw.push_back(-arc_weight[a]);
        tb.arc_uniq.push_back(arc_ilabel[a] - 1);  // pdf for now; replaced by the uniq id below
      }
    }
    tb.out_begin.push_back((int32_t)tb.arc_src.size() - abase);
    const int32_t narc = (int32_t)tb.arc_src.size() - abase;
    // in-arc lists (stable by arc id)
    {
      std::vector<int32_t> cnt(nloc + 1, 0);
      for (int32_t a = 0; a < narc; ++a) cnt[tb.arc_dst[abase + a] + 1]++;
      for (int32_t i = 0; i < nloc; ++i) cnt[i + 1] += cnt[i];
      for (int32_t i = 0; i <= nloc; ++i) tb.in_begin.push_back(cnt[i]);
      std::vector<int32_t> fillv(cnt.begin(), cnt.end() - 1);
      const size_t ib = tb.in_arc.size();
      tb.in_arc.resize(ib + narc);
      for (int32_t a = 0; a < narc; ++a) tb.in_arc[ib + fillv[tb.arc_dst[abase + a]]++] = a;
    }
    // unique (frame, pdf) pairs, in first-occurrence order like [K] ComputeLookupIndexes
    {
      const int32_t ubase = (int32_t)tb.uniq_t.size();
      // (counting sort by unique id: a list per id would be one heap allocation per (frame, pdf) pair -- tens of
      // thousands per minibatch, most of this function's time when a fresh supervision arrives every step)
      map_this_frame.assign(label_dim, -1);
      touched.clear();
      ucount.clear();
      int32_t cur_t = 0;
      int32_t t = 0;  // arcs are emitted in source-state order, so the frame only ever moves forward
      for (int32_t a = 0; a < narc; ++a) {
        const int32_t ls = tb.arc_src[abase + a];
        while (t + 1 <= T && tb.level_begin[(size_t)q * (T + 2) + t + 1] <= ls) ++t;
        if (t != cur_t) {
          for (int32_t p : touched) map_this_frame[p] = -1;
          touched.clear();
          cur_t = t;
        }
        const int32_t pdf = tb.arc_uniq[abase + a];
        int32_t u = map_this_frame[pdf];
        if (u < 0) {
          u = (int32_t)ucount.size();
          map_this_frame[pdf] = u;
          touched.push_back(pdf);
          tb.uniq_t.push_back(t);
          tb.uniq_pdf.push_back(pdf);
          ucount.push_back(0);
        }
        ucount[u]++;
        tb.arc_uniq[abase + a] = u;
      }
      int32_t run = 0;
      const size_t ub = tb.uniq_begin.size(), ua = tb.uniq_arc.size();
      for (int32_t c : ucount) {
        tb.uniq_begin.push_back(run);
        run += c;
      }
      tb.uniq_arc.resize(ua + narc);
      for (size_t u = 0; u < ucount.size(); ++u) ucount[u] = tb.uniq_begin[ub + u];  // fill positions
      for (int32_t a = 0; a < narc; ++a) tb.uniq_arc[ua + ucount[tb.arc_uniq[abase + a]]++] = a;
      tb.uniq_begin.push_back(run);
      tb.seq_uniq_off.push_back((int32_t)tb.uniq_t.size());
      tb.max_uniq = std::max(tb.max_uniq, (int32_t)tb.uniq_t.size() - ubase);
    }
    tb.seq_state_off.push_back((int32_t)tb.final_logw.size());
    tb.seq_arc_off.push_back((int32_t)tb.arc_src.size());
    tb.max_states = std::max(tb.max_states, nloc);
    tb.max_arcs = std::max(tb.max_arcs, narc);
  }
  *out = sup;
  return TC_OK;
}

void tc_supervision_free(tc_supervision *sup) {
  if (!sup) return;
  // the device tables go back to the per-device pool; they are reused once the work recorded on them is done
  for (auto &kv : sup->dev) pool_release(kv.first, kv.second.slot);
  delete sup;
}

int32_t tc_supervision_num_pdf(const tc_supervision *s) { return s ? s->P : 0; }
int32_t tc_supervision_num_sequence(const tc_supervision *s) { return s ? s->S : 0; }
int32_t tc_supervision_num_frame(const tc_supervision *s) { return s ? s->T : 0; }
float tc_supervision_weight(const tc_supervision *s) { return s ? s->weight : 0.f; }

// The host half of an upload: the supervision's tables into the pinned staging of a pool slot (2-3 MB of memcpy for 64
// x 150 frames).  No stream is involved, so a reader's look-ahead thread can do it (tc_supervision_stage) and the
// training thread's first use of the supervision only enqueues the copy.  Caller holds sup->mu.
static int stage_locked(tc_supervision *sup, int device) {
  if (sup->dev.count(device)) return TC_OK;
  NumTables &tb = sup->tab;
  struct Part { const void *src; size_t bytes; };
  const Part parts[] = {
      {tb.seq_state_off.data(), tb.seq_state_off.size() * 4}, {tb.seq_arc_off.data(), tb.seq_arc_off.size() * 4},
      {tb.seq_uniq_off.data(), tb.seq_uniq_off.size() * 4},   {tb.level_begin.data(), tb.level_begin.size() * 4},
      {tb.out_begin.data(), tb.out_begin.size() * 4},         {tb.in_begin.data(), tb.in_begin.size() * 4},
      {tb.in_arc.data(), tb.in_arc.size() * 4},               {tb.arc_src.data(), tb.arc_src.size() * 4},
      {tb.arc_dst.data(), tb.arc_dst.size() * 4},             {tb.arc_uniq.data(), tb.arc_uniq.size() * 4},
      {tb.uniq_t.data(), tb.uniq_t.size() * 4},               {tb.uniq_pdf.data(), tb.uniq_pdf.size() * 4},
      {tb.uniq_begin.data(), tb.uniq_begin.size() * 4},       {tb.uniq_arc.data(), tb.uniq_arc.size() * 4},
      {tb.arc_logw.data(), tb.arc_logw.size() * 4},           {tb.final_logw.data(), tb.final_logw.size() * 4},
  };
  const int nparts = (int)(sizeof(parts) / sizeof(parts[0]));
  size_t total = 0;
  sup->blob_off.clear();
  for (int i = 0; i < nparts; ++i) {
    sup->blob_off.push_back(total);
    total += (parts[i].bytes + 255) & ~(size_t)255;
  }
  total += 256;
  // behind the tables: one float per unique (frame, pdf), where the numerator kernel leaves its posteriors when it
  // runs beside the denominator (api.cpp); not part of the upload
  const size_t upload = total, stage_off = total;
  total += (tb.uniq_t.size() * 4 + 255) & ~(size_t)255;
  // A new supervision arrives with every minibatch: its tables live in a slot of a per-device pool (device
  // blob + PINNED host staging + two events), so that after warm-up a step neither allocates nor frees
  // device memory and the upload is a true asynchronous copy.
  PoolSlot *slot = nullptr;
  int rc = pool_acquire(device, total, &slot);
  if (rc != TC_OK) return rc;
  for (int i = 0; i < nparts; ++i)
    if (parts[i].bytes) memcpy(slot->host + sup->blob_off[i], parts[i].src, parts[i].bytes);
  NumDev d;
  d.slot = slot;
  d.upload_bytes = upload;
  d.uploaded = false;
  char *blob = slot->blob;
  auto P = [&](int i) { return (const int32_t *)(blob + sup->blob_off[i]); };
  d.seq_state_off = P(0); d.seq_arc_off = P(1); d.seq_uniq_off = P(2); d.level_begin = P(3);
  d.out_begin = P(4); d.in_begin = P(5); d.in_arc = P(6); d.arc_src = P(7); d.arc_dst = P(8);
  d.arc_uniq = P(9); d.uniq_t = P(10); d.uniq_pdf = P(11); d.uniq_begin = P(12); d.uniq_arc = P(13);
  d.arc_logw = (const float *)P(14);
  d.final_logw = (const float *)P(15);
  d.stage = (float *)(blob + stage_off);
  sup->dev[device] = d;
  return TC_OK;
}

int tc_supervision_stage(tc_supervision *sup, int device) {
  if (!sup) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sup->mu);
  return stage_locked(sup, device);
}

int tc_supervision_prepare(tc_supervision *sup, int device, void *stream_v) {
  if (!sup) return TC_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_v;
  std::lock_guard<std::mutex> lock(sup->mu);
  auto it = sup->dev.find(device);
  // A stream that is being captured into a graph can neither wait for nor ask about an event of this library: HIP counts
  // an event last recorded on that stream BEFORE the capture began as a captured one (hipStreamWaitEvent:
  // hipErrorStreamCaptureIsolation; hipEventSynchronize / hipEventQuery: hipErrorCapturedEvent, and the capture is
  // invalidated).  So inside a capture the upload is a node of the graph itself -- every replay copies the tables again
  // from the slot's own pinned block, ordered before the kernels that read them by the graph -- and no event is touched.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive;
  if (it != sup->dev.end() && it->second.uploaded && !capturing) {
    // uploaded earlier, possibly on another stream: order this stream behind the copy
    TC_HIP_CHECK(hipStreamWaitEvent(stream, it->second.slot->ready, 0));
    return TC_OK;
  }
  if (it == sup->dev.end()) {
    // (a pool that has to grow allocates: allowed during somebody's capture only in the relaxed mode)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (capturing) TC_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&mode));
    const int rc = stage_locked(sup, device);
    if (capturing) (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (rc != TC_OK) return rc;
    it = sup->dev.find(device);
  }
  PoolSlot *slot = it->second.slot;
  hipError_t e = hipMemcpyAsync(slot->blob, slot->host, it->second.upload_bytes, hipMemcpyHostToDevice, stream);
  if (capturing) {
    // (the supervision's state for calls outside the graph is left as it was)
    TC_HIP_CHECK(e);
    return TC_OK;
  }
  if (e == hipSuccess) e = hipEventRecord(slot->ready, stream);
  if (e == hipSuccess) e = hipEventRecord(slot->done, stream);  // (moved forward by every launch that reads the slot)
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    pool_release(device, slot);
    sup->dev.erase(it);
    return TC_ERR_HIP;
  }
  it->second.uploaded = true;
  return TC_OK;
}

}  // extern "C"
