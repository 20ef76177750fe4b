// Native random-access minibatch reader: what the reference exposes as my_lib_example_rand_reader_new / _reset /
// _num_batch / _num_data / _next / _free, my_lib_example_rand_feats and my_lib_supervision_rand_new (src/my_lib.h:8-17)
// over its RandReader (src/my_lib_example_rand.cpp:35-177): examples of equal frames_per_sequence are grouped into
// minibatches, shuffled inside and across the groups with a Mersenne twister seeded by the caller, and every minibatch
// is merged ([K] MergeChainExamples) when it is used.  Here additionally
//   * the reader is rank-aware: rank r of `world` ranks takes batches r, r + world, ... of the one shuffled list every
//     rank forms from the same seed, and every rank gets the same number of batches (the reference's only multi-GPU
//     recipe, example/chime5/parallel_train.py:26-75, splits one gathered batch on the host instead);
//   * the next `lookahead` batches are read, parsed, merged and their supervision handles built on worker threads while
//     the caller trains on the current one (round 3 did this in Python: 0.27 + 0.16 ms of interpreter time per step).
// Host code, with one exception the caller asks for: after tc_rand_reader_set_device the look-ahead threads also fill the
// pinned staging of each supervision's upload (tc_supervision_stage: hipSetDevice and the per-device pool's pinned /
// device allocations of supervision.cpp, made on those threads).  Without that call no GPU is touched here --
// tc_supervision_create builds host tables, which reach the device with the first loss call.
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "torchain_hip.h"

namespace {

struct Entry {
  std::string key, path;
  int64_t offset = -1;
};

struct Ready {
  tc_example *example = nullptr;
  tc_supervision *sup = nullptr;
  int rc = TC_OK;
  std::string error;
};

void release(Ready &r) {
  if (r.example) tc_example_free(r.example);
  if (r.sup) tc_supervision_free(r.sup);
  r = Ready();
}

thread_local std::string g_error;

}  // namespace

struct tc_rand_reader {
  std::vector<Entry> entries;
  std::map<int32_t, std::vector<int32_t>> by_length;  // frames_per_sequence -> entries (ascending lengths, scp order)
  // order == TC_RAND_ORDER_REFERENCE: the reference's own container, filled in the order it fills its own
  // (src/my_lib_example_rand.cpp:41,69-110): the lengths come out in libstdc++'s hash-table iteration order
  std::unordered_map<size_t, std::vector<int32_t>> by_length_ref;
  int order = TC_RAND_ORDER_SORTED;
  std::mt19937 engine;
  int batchsize = 1, rank = 0, world = 1, lookahead = 0;
  int device = -1;  // >= 0: look-ahead threads stage the supervisions they build for this GPU
  int64_t n_data = 0;
  std::vector<std::vector<int32_t>> batches;  // this rank's batches of the current epoch
  int64_t pos = -1;                            // the current batch (-1: before the first)
  Ready cur;
  // look-ahead: batch index -> result; workers take the lowest index not yet claimed
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::map<int64_t, Ready> done;
  int64_t next_claim = 0, epoch = 0, want_until = 0;
  bool stop = false;
  std::vector<std::thread> workers;

  // Fisher-Yates with the draw j = (engine() * (i + 1)) >> 32: std::shuffle's algorithm is not specified, so neither
  // this nor any other statement of it reproduces the reference's order; what is kept is its generator and structure.
  template <class V>
  void shuffle(V &v) {
    for (size_t i = v.size(); i > 1; --i) {
      const size_t j = (size_t)(((uint64_t)engine() * (uint64_t)i) >> 32);
      std::swap(v[i - 1], v[j]);
    }
  }

  void shuffle_keys() {  // reference: RandReader::shuffle_keys
    std::vector<std::vector<int32_t>> all;
    if (order == TC_RAND_ORDER_REFERENCE) {
      // The reference's statement on the reference's library calls (src/my_lib_example_rand.cpp:119-141): the lengths in the
      // iteration order of the std::unordered_map, each length's keys COPIED out of the map (`for (auto kv : ...)`: every
      // epoch starts from the file order again) and passed to std::shuffle, cut into batches, the batches passed to
      // std::shuffle.  What comes out is the order of the standard library this file is compiled with -- libstdc++ here
      // and in the reference's build.
      std::vector<int32_t> batch;
      batch.reserve((size_t)batchsize);
      for (auto kv : by_length_ref) {
        auto &keys = kv.second;
        std::shuffle(keys.begin(), keys.end(), engine);
        for (const auto &k : keys) {
          batch.push_back(k);
          if (batch.size() == (size_t)batchsize) {
            all.push_back(batch);
            batch.clear();
          }
        }
        if (batch.size() > 0) all.push_back(batch);
        batch.clear();
      }
      std::shuffle(all.begin(), all.end(), engine);
    } else {
      for (auto &kv : by_length) {
        std::vector<int32_t> keys = kv.second;
        shuffle(keys);
        for (size_t i = 0; i < keys.size(); i += (size_t)batchsize)
          all.emplace_back(keys.begin() + (long)i, keys.begin() + (long)std::min(keys.size(), i + (size_t)batchsize));
      }
      shuffle(all);
    }
    // this rank's share: the same number of batches on every rank (the last all.size() % world batches of the epoch's
    // list are left out; the list is shuffled anew every epoch)
    batches.clear();
    const size_t per_rank = all.size() / (size_t)world;
    for (size_t i = 0; i < per_rank; ++i) batches.push_back(std::move(all[i * (size_t)world + (size_t)rank]));
  }

  // (`dev`: the staging device as the caller read it under `mu`)
  Ready load(const std::vector<int32_t> &batch, int dev) const {
    Ready r;  // (filled in place: what an exception leaves half-built is released here, not lost with a local of the callee)
    try {
      load_into(r, batch, dev);
    } catch (const std::exception &e) {  // (bad_alloc and whatever else: a failed batch, not std::terminate on a worker)
      release(r);
      r.rc = TC_ERR_IO;
      r.error = std::string("reading a minibatch failed: ") + e.what();
    } catch (...) {
      release(r);
      r.rc = TC_ERR_IO;
      r.error = "reading a minibatch failed";
    }
    return r;
  }

  void load_into(Ready &r, const std::vector<int32_t> &batch, int dev) const {
    std::vector<const char *> paths;
    std::vector<int64_t> offs;
    for (int32_t e : batch) {
      paths.push_back(entries[(size_t)e].path.c_str());
      offs.push_back(entries[(size_t)e].offset);
    }
    r.rc = tc_example_read(paths.data(), offs.data(), (int32_t)paths.size(), 1, &r.example);
    if (r.rc != TC_OK) {
      r.error = tc_example_last_error();
      return;
    }
    const char *name = nullptr;
    int32_t nidx = 0, dims[5] = {0, 0, 0, 0, 0};
    const int32_t *idx = nullptr, *ab = nullptr, *il = nullptr, *nx = nullptr;
    const float *dw = nullptr, *aw = nullptr, *fin = nullptr;
    float weight = 0.f;
    r.rc = tc_example_output(r.example, 0, &name, &nidx, &idx, &dw, &weight, dims, &ab, &il, &aw, &nx, &fin);
    if (r.rc == TC_OK) r.rc = tc_supervision_create(&r.sup, weight, dims[0], dims[1], dims[2], dims[3], ab, il, aw, nx, fin);
    if (r.rc != TC_OK) r.error = "the minibatch's supervision does not build (tc_supervision_create)";
    if (r.rc == TC_OK && dev >= 0) (void)tc_supervision_stage(r.sup, dev);  // (a failure shows at the first use)
  }

  void worker() {
    std::unique_lock<std::mutex> lock(mu);
    for (;;) {
      cv_work.wait(lock, [&] { return stop || next_claim < want_until; });
      if (stop) return;
      const int64_t mine = next_claim++, my_epoch = epoch;
      Ready r;
      std::vector<int32_t> batch;
      const int dev = device;  // (tc_rand_reader_set_device writes it under this lock)
      try {
        batch = batches[(size_t)mine];
      } catch (...) {
        r.rc = TC_ERR_IO;
        r.error = "out of memory";
      }
      lock.unlock();
      if (r.rc == TC_OK) r = load(batch, dev);
      lock.lock();
      if (my_epoch == epoch && !stop)
        done[mine] = r;
      else
        release(r);  // (the reader was reset meanwhile)
      cv_done.notify_all();
    }
  }

  void drop_lookahead() {  // caller holds mu
    ++epoch;
    for (auto &kv : done) release(kv.second);
    done.clear();
    next_claim = want_until = 0;
  }

  ~tc_rand_reader() {
    {
      std::lock_guard<std::mutex> lock(mu);
      stop = true;
    }
    cv_work.notify_all();
    for (auto &t : workers) t.join();
    for (auto &kv : done) release(kv.second);
    release(cur);
  }
};

extern "C" {

const char *tc_rand_reader_last_error(void) { return g_error.c_str(); }

int tc_rand_reader_new(const char *scp_path, int seed, int batchsize, const char *len_file, int rank, int world,
                       int lookahead, tc_rand_reader **out) {
  return tc_rand_reader_new_ordered(scp_path, seed, batchsize, len_file, rank, world, lookahead, TC_RAND_ORDER_SORTED, out);
}

int tc_rand_reader_new_ordered(const char *scp_path, int seed, int batchsize, const char *len_file, int rank, int world,
                               int lookahead, int order, tc_rand_reader **out) {
  if (!scp_path || !out || batchsize <= 0 || world <= 0 || rank < 0 || rank >= world || lookahead < 0 || lookahead > 64 ||
      (order != TC_RAND_ORDER_SORTED && order != TC_RAND_ORDER_REFERENCE))
    return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  g_error.clear();
  try {
    std::unique_ptr<tc_rand_reader> r(new tc_rand_reader());
    r->engine.seed((uint32_t)seed);
    r->batchsize = batchsize;
    r->rank = rank;
    r->world = world;
    r->lookahead = lookahead;
    r->order = order;
    std::ifstream scp(scp_path);
    if (!scp.is_open()) {
      g_error = std::string("cannot open ") + scp_path;
      return TC_ERR_IO;
    }
    std::map<std::string, int32_t> index;
    std::string line;
    while (std::getline(scp, line)) {  // "key path:offset" or "key path"
      const size_t a = line.find_first_not_of(" \t\r");
      if (a == std::string::npos) continue;
      const size_t b = line.find_first_of(" \t", a);
      if (b == std::string::npos) continue;
      Entry e;
      e.key = line.substr(a, b - a);
      size_t c = line.find_first_not_of(" \t", b), d = line.find_last_not_of(" \t\r");
      if (c == std::string::npos) continue;
      std::string loc = line.substr(c, d - c + 1);
      const size_t colon = loc.rfind(':');
      if (colon != std::string::npos && colon + 1 < loc.size() &&
          loc.find_first_not_of("0123456789", colon + 1) == std::string::npos) {
        try {
          e.offset = std::stoll(loc.substr(colon + 1));
        } catch (const std::exception &) {  // (out_of_range: more digits than an offset has)
          g_error = "malformed scp line of key " + e.key + ": offset " + loc.substr(colon + 1);
          return TC_ERR_BAD_FST;
        }
        loc.resize(colon);
      }
      e.path = loc;
      index[e.key] = (int32_t)r->entries.size();
      r->entries.push_back(std::move(e));
    }
    // lengths: "key length" pairs of the length file (rspec + ".len" unless given), else from the examples themselves
    std::ifstream lens(len_file && *len_file ? std::string(len_file) : std::string(scp_path) + ".len");
    if (lens.is_open()) {
      std::string key;
      long length;
      while (lens >> key >> length) {
        auto it = index.find(key);
        if (it == index.end()) continue;  // (a key the scp does not hold cannot be read)
        r->by_length[(int32_t)length].push_back(it->second);
        r->by_length_ref[(size_t)length].push_back(it->second);
        ++r->n_data;
      }
    } else {
      for (size_t i = 0; i < r->entries.size(); ++i) {
        const char *path = r->entries[i].path.c_str();
        tc_example *ex = nullptr;
        int rc = tc_example_read(&path, &r->entries[i].offset, 1, 0, &ex);
        int32_t dims[5] = {0, 0, 0, 0, 0};
        if (rc == TC_OK) rc = tc_example_output(ex, 0, nullptr, nullptr, nullptr, nullptr, nullptr, dims, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (ex) tc_example_free(ex);
        if (rc != TC_OK) {
          g_error = "cannot read the example of key " + r->entries[i].key + ": " + tc_example_last_error();
          return rc;
        }
        r->by_length[dims[1]].push_back((int32_t)i);
        r->by_length_ref[(size_t)dims[1]].push_back((int32_t)i);
        ++r->n_data;
      }
    }
    r->shuffle_keys();
    for (int i = 0; i < lookahead; ++i) r->workers.emplace_back([p = r.get()] { p->worker(); });
    *out = r.release();
  } catch (...) {
    g_error = "out of memory";
    return TC_ERR_IO;
  }
  return TC_OK;
}

void tc_rand_reader_free(tc_rand_reader *r) { delete r; }

int tc_rand_reader_set_device(tc_rand_reader *r, int device) {
  if (!r) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(r->mu);
  r->device = device;
  return TC_OK;
}

int tc_rand_reader_reset(tc_rand_reader *r) {
  if (!r) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(r->mu);
  r->drop_lookahead();
  release(r->cur);
  r->pos = -1;
  r->shuffle_keys();
  return TC_OK;
}

int tc_rand_reader_num_batch(const tc_rand_reader *r) { return r ? (int)r->batches.size() : 0; }
int tc_rand_reader_num_data(const tc_rand_reader *r) { return r ? (int)r->n_data : 0; }

int tc_rand_reader_next(tc_rand_reader *r) {
  if (!r) return TC_ERR_INVALID_ARGUMENT;
  g_error.clear();
  std::unique_lock<std::mutex> lock(r->mu);
  release(r->cur);
  if (r->pos + 1 >= (int64_t)r->batches.size()) {
    r->pos = (int64_t)r->batches.size();
    return 0;
  }
  const int64_t want = ++r->pos;
  if (r->lookahead > 0) {
    r->want_until = std::min<int64_t>((int64_t)r->batches.size(), want + 1 + r->lookahead);
    r->cv_work.notify_all();
    r->cv_done.wait(lock, [&] { return r->done.count(want) != 0; });
    r->cur = r->done[want];
    r->done.erase(want);
  } else {
    const std::vector<int32_t> batch = r->batches[(size_t)want];
    const int dev = r->device;
    lock.unlock();
    Ready got = r->load(batch, dev);
    lock.lock();
    r->cur = got;
  }
  if (r->cur.rc != TC_OK) {
    g_error = r->cur.error;
    return r->cur.rc;
  }
  return 1;
}

int tc_rand_reader_example(const tc_rand_reader *r, const tc_example **out) {
  if (!r || !out || !r->cur.example) return TC_ERR_INVALID_ARGUMENT;
  *out = r->cur.example;
  return TC_OK;
}

int tc_rand_reader_take_example(tc_rand_reader *r, tc_example **out) {
  if (!r || !out || !r->cur.example) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(r->mu);
  *out = r->cur.example;  // the caller's from here on (tc_example_free)
  r->cur.example = nullptr;
  return TC_OK;
}

int tc_rand_reader_supervision_new(tc_rand_reader *r, tc_supervision **out) {
  if (!r || !out || (!r->cur.example && !r->cur.sup)) return TC_ERR_INVALID_ARGUMENT;
  std::unique_lock<std::mutex> lock(r->mu);
  if (r->cur.sup) {  // the handle the look-ahead built: handed over
    *out = r->cur.sup;
    r->cur.sup = nullptr;
    return TC_OK;
  }
  lock.unlock();
  if (!r->cur.example) return TC_ERR_INVALID_ARGUMENT;  // (both already taken)
  int32_t dims[5] = {0, 0, 0, 0, 0};
  const int32_t *ab = nullptr, *il = nullptr, *nx = nullptr;
  const float *aw = nullptr, *fin = nullptr;
  float weight = 0.f;
  int rc = tc_example_output(r->cur.example, 0, nullptr, nullptr, nullptr, nullptr, &weight, dims, &ab, &il, &aw, &nx, &fin);
  if (rc == TC_OK) rc = tc_supervision_create(out, weight, dims[0], dims[1], dims[2], dims[3], ab, il, aw, nx, fin);
  return rc;
}

int tc_rand_reader_batch_keys(const tc_rand_reader *r, int32_t batch, char *buf, int32_t cap) {
  if (!r || batch < 0 || batch >= (int32_t)r->batches.size() || !buf || cap <= 0) return TC_ERR_INVALID_ARGUMENT;
  std::string s;
  for (int32_t e : r->batches[(size_t)batch]) {
    if (!s.empty()) s += ' ';
    s += r->entries[(size_t)e].key;
  }
  if ((int32_t)s.size() + 1 > cap) return TC_ERR_WORKSPACE;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)r->batches[(size_t)batch].size();
}

}  // extern "C"
