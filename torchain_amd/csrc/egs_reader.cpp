// Native reader of chain-egs minibatches: the data format on the caller's side of the hot path.
//
// Replaces what the reference reaches through Kaldi in src/my_lib_example_rand.cpp:35-177 (RandomAccess reader of
// NnetChainExample + kaldi::nnet3::MergeChainExamples): tc_example_read opens the scp entries of one minibatch, parses
// the binary <Nnet3ChainEg> objects and merges them -- inputs stacked example by example with n = position in the
// batch, supervisions appended (supervision_merge.cpp), output indexes and deriv_weights frame-major.  Host memory
// only; called without the interpreter lock, so RandExample's look-ahead thread really runs beside the training step.
// (torchain_amd/egs.py holds the same format knowledge in numpy and is what the tests check this against.)
//
// Formats, restated from Kaldi's / OpenFst's published sources (base/io-funcs-inl.h, nnet3/nnet-common.cc
// WriteIndexVector, matrix/compressed-matrix.cc, nnet3/nnet-chain-example.cc, chain/chain-supervision.cc; OpenFst
// compact-fst.h) -- see egs.py's header for the grammar.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "chain_internal.h"

namespace {

struct FormatError {
  const char *what;
};

// Buffered reader over a FILE (fread keeps its own buffer; examples are a few tens of KB each)
struct In {
  std::FILE *f;
  void read(void *dst, size_t n) {
    if (n && fread_unlocked(dst, 1, n, f) != n) throw FormatError{"unexpected end of file"};
  }
  // n elements into v, growing v as the data arrives: a size field in a damaged or crafted header can make the reader
  // allocate no more than the bytes that are really there (+ one step)
  template <class T>
  void read_grow(std::vector<T> *v, size_t n) {
    const size_t step_max = ((size_t)16 << 20) / sizeof(T);
    v->clear();
    for (size_t done = 0; done < n;) {
      const size_t step = n - done < step_max ? n - done : step_max;
      v->resize(done + step);
      read(v->data() + done, step * sizeof(T));
      done += step;
    }
  }
  int peek() {
    const int c = getc_unlocked(f);
    if (c != EOF) std::ungetc(c, f);
    return c;
  }
  uint8_t byte() {  // (the FILE belongs to this call alone: no per-byte locking)
    const int c = getc_unlocked(f);
    if (c == EOF) throw FormatError{"unexpected end of file"};
    return (uint8_t)c;
  }
  std::string token() {
    std::string out;
    for (;;) {
      const int c = getc_unlocked(f);
      if (c == EOF) throw FormatError{"unexpected end of file in a token"};
      if (c == ' ' || c == '\t' || c == '\n') {
        if (!out.empty()) return out;
        continue;
      }
      out.push_back((char)c);
      if (out.size() > 256) throw FormatError{"token too long"};
    }
  }
  void expect(const char *tok) {
    if (token() != tok) throw FormatError{tok};
  }
  template <class T>
  T basic() {  // [K] ReadBasicType, binary: one size byte, then the value
    if (byte() != sizeof(T)) throw FormatError{"basic type of unexpected size"};
    T v;
    read(&v, sizeof(T));
    return v;
  }
  bool boolean() {
    const uint8_t c = byte();
    if (c != 'T' && c != 'F') throw FormatError{"bad bool"};
    if (peek() == ' ') byte();
    return c == 'T';
  }
};

struct Io {
  std::string name;
  std::vector<int32_t> idx;  // (n, t, x) per index
  int32_t rows = 0, cols = 0;
  std::vector<float> feat;
};

struct Sup {
  float weight = 0.f;
  int32_t S = 0, T = 0, P = 0, nstates = 0;
  std::vector<int32_t> ab, il, nx;
  std::vector<float> w, fin;
};

struct Out {
  std::string name;
  std::vector<int32_t> idx;
  Sup sup;
  std::vector<float> dw;
};

void read_index_vector(In &in, std::vector<int32_t> *out) {
  in.expect("<I1V>");
  const int32_t size = in.basic<int32_t>();
  if (size < 0 || size > (1 << 26)) throw FormatError{"bad index vector size"};
  out->clear();
  out->reserve((size_t)(size < (1 << 20) ? size : (1 << 20)) * 3);  // (grows with the elements that are really there)
  int32_t n = 0, t = 0, x = 0;
  for (int32_t i = 0; i < size; ++i) {
    const int8_t c = (int8_t)in.byte();
    if (c > -125 && c < 125) {
      if (i == 0) {
        n = 0;
        t = c;
        x = 0;
      } else {
        t += c;
      }
    } else {
      if (c != 127) throw FormatError{"bad index vector element"};
      n = in.basic<int32_t>();
      t = in.basic<int32_t>();
      x = in.basic<int32_t>();
    }
    out->push_back(n);
    out->push_back(t);
    out->push_back(x);
  }
}

void read_general_matrix(In &in, Io *io) {
  const std::string tok = in.token();
  // (the raw bytes are read first, in steps: the float matrix is allocated for data that exists)
  auto dims = [&](int32_t rows, int32_t cols) {
    if (rows < 0 || cols < 0 || (int64_t)rows * cols > ((int64_t)1 << 31)) throw FormatError{"bad matrix size"};
    io->rows = rows;
    io->cols = cols;
    return (size_t)rows * cols;
  };
  if (tok == "FM" || tok == "DM") {
    const int32_t rows = in.basic<int32_t>(), cols = in.basic<int32_t>();
    const size_t n = dims(rows, cols);
    if (tok == "FM") {
      in.read_grow(&io->feat, n);
    } else {
      std::vector<double> d;
      in.read_grow(&d, n);
      io->feat.resize(n);
      for (size_t i = 0; i < n; ++i) io->feat[i] = (float)d[i];
    }
    return;
  }
  if (tok == "CM" || tok == "CM2" || tok == "CM3") {
    struct {
      float min_value, range;
      int32_t rows, cols;
    } h;
    in.read(&h, 16);
    const size_t n = dims(h.rows, h.cols);
    if (tok == "CM") {
      // per column four uint16 percentiles (0, 25, 75, 100), then column-major bytes
      std::vector<uint16_t> hdr;
      in.read_grow(&hdr, (size_t)4 * h.cols);
      std::vector<uint8_t> b;
      in.read_grow(&b, n);
      io->feat.resize(n);
      for (int32_t c = 0; c < h.cols; ++c) {
        float p[4];
        for (int k = 0; k < 4; ++k) p[k] = h.min_value + h.range * (float)hdr[4 * (size_t)c + k] / 65535.0f;
        for (int32_t r = 0; r < h.rows; ++r) {
          const float v = (float)b[(size_t)c * h.rows + r];
          float y;
          if (v <= 64.f)
            y = p[0] + (p[1] - p[0]) * v * (1.0f / 64.0f);
          else if (v <= 192.f)
            y = p[1] + (p[2] - p[1]) * (v - 64.0f) * (1.0f / 128.0f);
          else
            y = p[2] + (p[3] - p[2]) * (v - 192.0f) * (1.0f / 63.0f);
          io->feat[(size_t)r * h.cols + c] = y;
        }
      }
    } else if (tok == "CM2") {
      std::vector<uint16_t> u;
      in.read_grow(&u, n);
      io->feat.resize(n);
      for (size_t i = 0; i < n; ++i) io->feat[i] = h.min_value + h.range * (float)u[i] / 65535.0f;
    } else {
      std::vector<uint8_t> u;
      in.read_grow(&u, n);
      io->feat.resize(n);
      for (size_t i = 0; i < n; ++i) io->feat[i] = h.min_value + h.range * (float)u[i] / 255.0f;
    }
    return;
  }
  throw FormatError{"unsupported matrix type (sparse features are not used by chain egs)"};
}

// OpenFst CompactFst<StdArc, AcceptorCompactor> as [K] Supervision::Write stores the numerator FST
void read_compact_acceptor(In &in, Sup *s) {
  int32_t magic;
  in.read(&magic, 4);
  if (magic != 2125659606) throw FormatError{"bad FST magic"};
  auto fst_string = [&]() {
    int32_t n;
    in.read(&n, 4);
    if (n < 0 || n > (1 << 16)) throw FormatError{"bad FST header string"};
    std::string v((size_t)n, '\0');
    in.read(&v[0], (size_t)n);
    return v;
  };
  const std::string fsttype = fst_string(), arctype = fst_string();
  int32_t version, flags;
  in.read(&version, 4);
  in.read(&flags, 4);
  uint64_t props;
  int64_t start, nstates, narcs;
  in.read(&props, 8);
  in.read(&start, 8);
  in.read(&nstates, 8);
  in.read(&narcs, 8);
  if (fsttype != "compact_acceptor" || arctype != "standard" || (flags & 4) || (start != 0 && start != -1))
    throw FormatError{"supervision FST must be an unaligned compact_acceptor over StdArc starting at state 0"};
  if (flags & 3) throw FormatError{"symbol tables inside a supervision FST are not supported"};
  if (nstates < 0 || nstates > (1 << 28)) throw FormatError{"bad FST state count"};
  std::vector<uint32_t> states;
  in.read_grow(&states, (size_t)nstates + 1);
  const size_t ncomp = nstates > 0 ? states[(size_t)nstates] : 0;
  if (ncomp > ((size_t)1 << 30)) throw FormatError{"bad FST element count"};
  struct Elem {
    int32_t label;
    float weight;
    int32_t next;
  };
  std::vector<Elem> comp;
  in.read_grow(&comp, ncomp);
  s->nstates = (int32_t)nstates;
  s->fin.assign((size_t)nstates, std::numeric_limits<float>::infinity());
  s->ab.assign((size_t)nstates + 1, 0);
  s->il.clear();
  s->w.clear();
  s->nx.clear();
  s->il.reserve(ncomp);
  s->w.reserve(ncomp);
  s->nx.reserve(ncomp);
  for (int64_t st = 0; st < nstates; ++st) {
    if (states[(size_t)st] > states[(size_t)st + 1] || states[(size_t)st + 1] > ncomp) throw FormatError{"bad FST offsets"};
    for (size_t e = states[(size_t)st]; e < states[(size_t)st + 1]; ++e) {
      if (comp[e].label == -1) {
        s->fin[(size_t)st] = comp[e].weight;
      } else {
        s->il.push_back(comp[e].label);
        s->w.push_back(comp[e].weight);
        s->nx.push_back(comp[e].next);
      }
    }
    s->ab[(size_t)st + 1] = (int32_t)s->il.size();
  }
}

void read_supervision(In &in, Sup *s) {
  in.expect("<Supervision>");
  in.expect("<Weight>");
  s->weight = in.basic<float>();
  in.expect("<NumSequences>");
  s->S = in.basic<int32_t>();
  in.expect("<FramesPerSeq>");
  s->T = in.basic<int32_t>();
  in.expect("<LabelDim>");
  s->P = in.basic<int32_t>();
  if (s->S < 1 || s->T < 1 || s->P < 1 || (int64_t)s->S * s->T > ((int64_t)1 << 26) || !(s->weight == s->weight))
    throw FormatError{"bad supervision dimensions"};
  if (in.peek() == '<') {  // later Kaldi: <End2End> flag (the FST otherwise starts with its magic number, never '<')
    in.expect("<End2End>");
    if (in.boolean()) throw FormatError{"end-to-end (e2e) supervisions are outside this path"};
  }
  read_compact_acceptor(in, s);
  // later Kaldi: alignment pdfs of the supervision, written behind the FST when the vector is not empty
  // ([K] Supervision::Write: WriteToken("<AlignmentPdfs>"), WriteIntegerVector).  The path does not use them: skipped.
  std::string tok = in.token();
  if (tok == "<AlignmentPdfs>") {
    if (in.byte() != sizeof(int32_t)) throw FormatError{"<AlignmentPdfs>: element size"};
    int32_t n = 0;
    in.read(&n, sizeof(n));
    if (n < 0 || (int64_t)n > (int64_t)s->S * s->T) throw FormatError{"<AlignmentPdfs>: size"};
    std::vector<int32_t> skip((size_t)n);
    in.read(skip.data(), (size_t)n * sizeof(int32_t));
    tok = in.token();
  }
  if (tok != "</Supervision>") throw FormatError{"</Supervision>"};
}

struct Example {
  std::vector<Io> in;
  std::vector<Out> out;
};

void read_example(In &in, Example *eg) {
  in.expect("<Nnet3ChainEg>");
  in.expect("<NumInputs>");
  const int32_t ni = in.basic<int32_t>();
  if (ni < 0 || ni > 64) throw FormatError{"bad number of inputs"};
  eg->in.resize((size_t)ni);
  for (auto &io : eg->in) {
    in.expect("<NnetIo>");
    io.name = in.token();
    read_index_vector(in, &io.idx);
    read_general_matrix(in, &io);
    in.expect("</NnetIo>");
  }
  in.expect("<NumOutputs>");
  const int32_t no = in.basic<int32_t>();
  if (no < 0 || no > 64) throw FormatError{"bad number of outputs"};
  eg->out.resize((size_t)no);
  for (auto &o : eg->out) {
    in.expect("<NnetChainSup>");
    o.name = in.token();
    read_index_vector(in, &o.idx);
    read_supervision(in, &o.sup);
    const std::string tok = in.token();
    if (tok == "<DW>") {  // [K] WriteVectorAsChar: size byte, int32 count, bytes scaled by 255
      if (in.byte() != 1) throw FormatError{"bad <DW> vector"};
      int32_t n;
      in.read(&n, 4);
      if (n < 0 || n > (1 << 26)) throw FormatError{"bad <DW> size"};
      std::vector<uint8_t> b;
      in.read_grow(&b, (size_t)n);
      o.dw.resize((size_t)n);
      for (int32_t i = 0; i < n; ++i) o.dw[(size_t)i] = (float)b[(size_t)i] / 255.0f;
      in.expect("</NnetChainSup>");
    } else if (tok == "<DW2>") {
      in.expect("FV");
      const int32_t n = in.basic<int32_t>();
      if (n < 0 || n > (1 << 26)) throw FormatError{"bad <DW2> size"};
      in.read_grow(&o.dw, (size_t)n);
      in.expect("</NnetChainSup>");
    } else if (tok == "</NnetChainSup>") {
      o.dw.assign(o.idx.size() / 3, 1.0f);
    } else {
      throw FormatError{"unexpected token in <NnetChainSup>"};
    }
  }
  in.expect("</Nnet3ChainEg>");
}

// [K] MergeChainExamples for examples with one output (egs.py: merge_chain_examples)
int merge(const std::vector<Example> &egs, Example *m) {
  const Example &first = egs[0];
  if (first.out.empty()) return TC_ERR_BAD_FST;
  // one output per example is what the path (and the reference: outputs[0], src/my_lib_example.cpp:75) handles: more are
  // refused rather than silently dropped; an input's rows must be the rows its indexes name
  for (const Example &eg : egs) {
    if (eg.out.size() != 1) return TC_ERR_INVALID_ARGUMENT;
    for (const Io &io : eg.in)
      if ((size_t)io.rows != io.idx.size() / 3) return TC_ERR_BAD_FST;
  }
  m->in.resize(first.in.size());
  for (size_t j = 0; j < first.in.size(); ++j) {
    Io &dst = m->in[j];
    dst.name = first.in[j].name;
    dst.cols = first.in[j].cols;
    size_t rows = 0, nidx = 0;
    for (const Example &eg : egs) {
      if (eg.in.size() != first.in.size() || eg.in[j].name != dst.name || eg.in[j].cols != dst.cols)
        return TC_ERR_INVALID_ARGUMENT;  // examples disagree on their inputs
      rows += (size_t)eg.in[j].rows;
      nidx += eg.in[j].idx.size();
    }
    dst.rows = (int32_t)rows;
    dst.feat.reserve(rows * (size_t)dst.cols);  // (appended, not resized: no zero-fill of megabytes that are then overwritten)
    dst.idx.reserve(nidx);
    for (size_t n = 0; n < egs.size(); ++n) {
      const Io &src = egs[n].in[j];
      dst.feat.insert(dst.feat.end(), src.feat.begin(), src.feat.end());
      for (size_t i = 0; i < src.idx.size(); i += 3) {
        dst.idx.push_back((int32_t)n);
        dst.idx.push_back(src.idx[i + 1]);
        dst.idx.push_back(src.idx[i + 2]);
      }
    }
  }
  // supervisions: AppendSupervision through the library's own entry point
  const Sup &s0 = first.out[0].sup;
  const int K = (int)egs.size();
  std::vector<int32_t> nst((size_t)K), frames((size_t)K), ab, il, nx;
  std::vector<float> aw, fin;
  int64_t cap_states = 0, cap_arcs = 0, total_S = 0;
  for (int k = 0; k < K; ++k) {
    if (egs[(size_t)k].out.empty()) return TC_ERR_INVALID_ARGUMENT;
    const Sup &s = egs[(size_t)k].out[0].sup;
    if (s.weight != s0.weight || s.T != s0.T || s.P != s0.P) return TC_ERR_INVALID_ARGUMENT;
    nst[(size_t)k] = s.nstates;
    frames[(size_t)k] = s.S * s.T;
    total_S += s.S;
    ab.insert(ab.end(), s.ab.begin(), s.ab.end());
    il.insert(il.end(), s.il.begin(), s.il.end());
    nx.insert(nx.end(), s.nx.begin(), s.nx.end());
    aw.insert(aw.end(), s.w.begin(), s.w.end());
    fin.insert(fin.end(), s.fin.begin(), s.fin.end());
    cap_states += s.nstates;
    cap_arcs += (int64_t)s.il.size();
    if (k > 0) {
      const Sup &prev = egs[(size_t)k - 1].out[0].sup;
      int64_t nfin = 0;
      for (float f : prev.fin) nfin += std::isinf(f) ? 0 : 1;
      cap_arcs += nfin * (s.nstates > 0 ? s.ab[1] : 0);
    }
  }
  Out &o = (m->out.resize(1), m->out[0]);
  o.name = first.out[0].name;
  Sup &ms = o.sup;
  ms.weight = s0.weight;
  ms.S = (int32_t)total_S;
  ms.T = s0.T;
  ms.P = s0.P;
  if (K == 1) {
    ms = s0;
  } else {
    ms.ab.resize((size_t)cap_states + 1);
    ms.il.resize((size_t)cap_arcs);
    ms.w.resize((size_t)cap_arcs);
    ms.nx.resize((size_t)cap_arcs);
    ms.fin.resize((size_t)cap_states);
    int32_t ns = 0;
    int64_t na = 0;
    const int rc = tc_supervision_append(K, nst.data(), frames.data(), ab.data(), il.data(), aw.data(), nx.data(), fin.data(),
                                         cap_states, cap_arcs, &ns, &na, ms.ab.data(), ms.il.data(), ms.w.data(),
                                         ms.nx.data(), ms.fin.data());
    if (rc != TC_OK) return rc;
    ms.nstates = ns;
    ms.ab.resize((size_t)ns + 1);
    ms.il.resize((size_t)na);
    ms.w.resize((size_t)na);
    ms.nx.resize((size_t)na);
    ms.fin.resize((size_t)ns);
  }
  // output indexes and deriv_weights: frame-major over the merged sequences
  const int T = s0.T;
  o.idx.resize((size_t)T * total_S * 3);
  o.dw.resize((size_t)T * total_S);
  int64_t base = 0;
  for (const Example &eg : egs) {
    const Out &src = eg.out[0];
    const int So = src.sup.S;
    if ((int64_t)src.idx.size() != (int64_t)3 * T * So || (int64_t)src.dw.size() != (int64_t)T * So)
      return TC_ERR_INVALID_ARGUMENT;
    for (int t = 0; t < T; ++t)
      for (int s = 0; s < So; ++s) {
        const size_t from = (size_t)t * So + s, to = (size_t)t * total_S + base + s;
        o.idx[3 * to] = (int32_t)(base + s);
        o.idx[3 * to + 1] = src.idx[3 * from + 1];
        o.idx[3 * to + 2] = src.idx[3 * from + 2];
        o.dw[to] = src.dw[from];
      }
    base += So;
  }
  return TC_OK;
}

thread_local std::string g_example_error;

}  // namespace

struct tc_example {
  Example eg;
};

namespace {
// A minibatch is ~5 MB of vectors: allocated fresh for every batch they come from mmap and are paged in (and zeroed by
// the kernel) while they are filled -- a third of the reader's time.  Freed examples wait here, emptied but with their
// capacity, for the next read (a handful: the batches a training loop has in flight).
constexpr size_t kExamplePoolMax = 12;
// The pool has PROCESS lifetime (allocated once, never destroyed): look-ahead threads of a reader that is still alive when
// the process exits (csrc/rand_reader.cpp) may be inside pool_take / tc_example_free while static destructors run, and a
// pool destroyed under them would leave them with a dead mutex.  A few MB of capacity the OS reclaims with the process.
struct ExamplePool {
  std::mutex mu;
  std::vector<tc_example *> v;
};
ExamplePool &examples() {
  static ExamplePool *const pool = new ExamplePool();
  return *pool;
}

void empty_example(Example *eg) {
  for (Io &io : eg->in) {
    io.name.clear();
    io.idx.clear();
    io.feat.clear();
    io.rows = io.cols = 0;
  }
  for (Out &o : eg->out) {
    o.name.clear();
    o.idx.clear();
    o.dw.clear();
    o.sup.ab.clear();
    o.sup.il.clear();
    o.sup.nx.clear();
    o.sup.w.clear();
    o.sup.fin.clear();
    o.sup.nstates = 0;
  }
}

tc_example *pool_take() {
  ExamplePool &pool = examples();
  std::lock_guard<std::mutex> lock(pool.mu);
  if (pool.v.empty()) return nullptr;
  tc_example *ex = pool.v.back();
  pool.v.pop_back();
  return ex;
}
}  // namespace

extern "C" {

int tc_example_read(const char *const *paths, const int64_t *offsets, int32_t n, int merge_single, tc_example **out) {
  if (!paths || !out || n <= 0) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  g_example_error.clear();
  // (per-thread scratch for the batch's examples, kept between calls: their vectors are refilled, not reallocated; the
  // entries of a batch mostly name one archive: one FILE while consecutive entries do)
  static thread_local std::vector<Example> egs;
  egs.resize((size_t)n);
  struct Closer {
    std::FILE *f = nullptr;
    ~Closer() {
      if (f) std::fclose(f);
    }
  } open;
  const char *open_path = nullptr;
  try {
    for (int32_t i = 0; i < n; ++i) {
      // (an entry without an offset reads from the start of a freshly opened file, which may not be seekable)
      if (!(open.f && open_path && paths[i] && offsets && offsets[i] >= 0 && std::strcmp(open_path, paths[i]) == 0)) {
        if (open.f) std::fclose(open.f);
        open.f = paths[i] ? std::fopen(paths[i], "rb") : nullptr;
        open_path = paths[i];
      }
      std::FILE *f = open.f;
      if (!f) {
        g_example_error = std::string("cannot open ") + (paths[i] ? paths[i] : "(null)");
        return TC_ERR_IO;
      }
      if (offsets && offsets[i] >= 0 && fseeko(f, (off_t)offsets[i], SEEK_SET) != 0) {
        g_example_error = "cannot seek";
        return TC_ERR_IO;
      }
      In in{f};
      char marker[2];
      in.read(marker, 2);
      if (marker[0] != '\0' || marker[1] != 'B') throw FormatError{"text-mode egs are not supported (expected \\0B)"};
      read_example(in, &egs[(size_t)i]);
    }
  } catch (const FormatError &e) {
    g_example_error = e.what;
    return TC_ERR_BAD_FST;
  } catch (...) {
    g_example_error = "out of memory";
    return TC_ERR_IO;
  }
  try {
    std::unique_ptr<tc_example> ex(pool_take());  // (owned here until it is handed out: a throwing merge leaks nothing)
    if (!ex) ex.reset(new tc_example());
    if (n == 1 && !merge_single) {
      ex->eg = egs[0];  // (a copy: the scratch keeps its buffers)
    } else {
      const int rc = merge(egs, &ex->eg);
      if (rc != TC_OK) {
        g_example_error = "examples of a minibatch do not merge (inputs, weight, frames or label-dim differ, or a "
                          "supervision is not a connected acceptor of the stated length)";
        return rc;
      }
    }
    *out = ex.release();
  } catch (...) {
    g_example_error = "out of memory";
    return TC_ERR_IO;
  }
  return TC_OK;
}

void tc_example_free(tc_example *ex) {
  if (!ex) return;
  empty_example(&ex->eg);
  {
    ExamplePool &pool = examples();
    std::lock_guard<std::mutex> lock(pool.mu);
    if (pool.v.size() < kExamplePoolMax) {
      pool.v.push_back(ex);
      return;
    }
  }
  delete ex;
}

// ---- sequential archives ("ark:file", "ark:command |"): key SPACE \0B object, one after the other
struct tc_archive {
  std::FILE *f = nullptr;
  bool pipe = false;
};

int tc_archive_open(const char *rxfilename, tc_archive **out) {
  if (!rxfilename || !out) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::string name(rxfilename);
  while (!name.empty() && (name.back() == ' ' || name.back() == '\t' || name.back() == '\n')) name.pop_back();
  const bool pipe = !name.empty() && name.back() == '|';
  if (pipe) name.pop_back();
  std::FILE *f = pipe ? popen(name.c_str(), "r") : std::fopen(name.c_str(), "rb");
  if (!f) {
    g_example_error = std::string("cannot open ") + rxfilename;
    return TC_ERR_IO;
  }
  tc_archive *a = new tc_archive();
  a->f = f;
  a->pipe = pipe;
  *out = a;
  return TC_OK;
}

int tc_archive_close(tc_archive *a) {
  if (!a) return TC_OK;
  int rc = TC_OK;
  if (a->f) {
    if (a->pipe) {
      if (pclose(a->f) != 0) rc = TC_ERR_IO;
    } else {
      std::fclose(a->f);
    }
  }
  delete a;
  return rc;
}

// 1: an example was read (its key in key[0..key_cap), NUL-terminated); 0: end of the archive; < 0: error
int tc_archive_next(tc_archive *a, char *key, int32_t key_cap, tc_example **out) {
  if (!a || !a->f || !key || key_cap < 2 || !out) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  g_example_error.clear();
  int c = getc_unlocked(a->f);
  while (c == ' ' || c == '\n') c = getc_unlocked(a->f);
  if (c == EOF) return 0;
  int32_t n = 0;
  while (c != ' ') {
    if (c == EOF || n + 1 >= key_cap) {
      g_example_error = c == EOF ? "archive ends inside a key" : "key too long";
      return TC_ERR_BAD_FST;
    }
    key[n++] = (char)c;
    c = getc_unlocked(a->f);
  }
  key[n] = '\0';
  try {
    tc_example *ex = new tc_example();
    try {
      In in{a->f};
      char marker[2];
      in.read(marker, 2);
      if (marker[0] != '\0' || marker[1] != 'B') throw FormatError{"text-mode egs are not supported (expected \\0B)"};
      read_example(in, &ex->eg);
    } catch (...) {
      delete ex;
      throw;
    }
    *out = ex;
  } catch (const FormatError &e) {
    g_example_error = e.what;
    return TC_ERR_BAD_FST;
  } catch (...) {
    g_example_error = "out of memory";
    return TC_ERR_IO;
  }
  return 1;
}

const char *tc_example_last_error(void) { return g_example_error.c_str(); }

int tc_example_counts(const tc_example *ex, int32_t *out2) {
  if (!ex || !out2) return TC_ERR_INVALID_ARGUMENT;
  out2[0] = (int32_t)ex->eg.in.size();
  out2[1] = (int32_t)ex->eg.out.size();
  return TC_OK;
}

int tc_example_input(const tc_example *ex, int32_t j, const char **name, int32_t *rows, int32_t *cols,
                     int32_t *num_indexes, const float **features, const int32_t **indexes) {
  if (!ex || j < 0 || j >= (int32_t)ex->eg.in.size()) return TC_ERR_INVALID_ARGUMENT;
  const Io &io = ex->eg.in[(size_t)j];
  if (name) *name = io.name.c_str();
  if (rows) *rows = io.rows;
  if (cols) *cols = io.cols;
  if (num_indexes) *num_indexes = (int32_t)(io.idx.size() / 3);
  if (features) *features = io.feat.data();
  if (indexes) *indexes = io.idx.data();
  return TC_OK;
}

int tc_example_output(const tc_example *ex, int32_t j, const char **name, int32_t *num_indexes, const int32_t **indexes,
                      const float **deriv_weights, float *weight, int32_t *dims5, const int32_t **arc_begin,
                      const int32_t **ilabel, const float **arc_weight, const int32_t **nextstate,
                      const float **final_weight) {
  if (!ex || j < 0 || j >= (int32_t)ex->eg.out.size()) return TC_ERR_INVALID_ARGUMENT;
  const Out &o = ex->eg.out[(size_t)j];
  if (name) *name = o.name.c_str();
  if (num_indexes) *num_indexes = (int32_t)(o.idx.size() / 3);
  if (indexes) *indexes = o.idx.data();
  if (deriv_weights) *deriv_weights = o.dw.data();
  if (weight) *weight = o.sup.weight;
  if (dims5) {
    dims5[0] = o.sup.S;
    dims5[1] = o.sup.T;
    dims5[2] = o.sup.P;
    dims5[3] = o.sup.nstates;
    dims5[4] = (int32_t)o.sup.il.size();
  }
  if (arc_begin) *arc_begin = o.sup.ab.data();
  if (ilabel) *ilabel = o.sup.il.data();
  if (arc_weight) *arc_weight = o.sup.w.data();
  if (nextstate) *nextstate = o.sup.nx.data();
  if (final_weight) *final_weight = o.sup.fin.data();
  return TC_OK;
}

}  // extern "C"
