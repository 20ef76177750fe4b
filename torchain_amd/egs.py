"""Kaldi chain examples ("egs") without Kaldi: the data format on the caller's side of the hot path
(SURVEY.md section 8f-3).

The reference reads ``<Nnet3ChainEg>`` archives through Kaldi's table readers and merges examples with
``kaldi::nnet3::MergeChainExamples`` (``src/my_lib_example.cpp:35-127``, ``src/my_lib_example_rand.cpp:35-177``);
``torchain/io.py:60-175`` wraps that as ``Example`` / ``RandExample`` / ``open_example``.  Kaldi is not a
dependency here, so this module parses the binary formats itself -- host-side numpy, nothing here runs on the GPU:

* table formats: ``ark:file``, ``ark,bg:file`` (sequential archives: ``key SPACE \\0B object``), ``scp:file``
  (``key path:offset`` lines) and bare paths;
* ``NnetChainExample`` = ``<Nnet3ChainEg> <NumInputs> n NnetIo* <NumOutputs> m NnetChainSupervision* </Nnet3ChainEg>``;
* ``NnetIo`` = name, index vector (``<I1V>``, delta-coded ``(n, t, x)``), ``GeneralMatrix`` (``FM``, ``DM``, compressed
  ``CM`` / ``CM2`` / ``CM3``);
* ``NnetChainSupervision`` = name, index vector, ``chain::Supervision`` (weight, num-sequences, frames-per-sequence,
  label-dim, FST as OpenFst ``compact_acceptor``), ``deriv_weights`` (``<DW>`` bytes or ``<DW2>`` floats);
* merging: features stacked example by example, supervisions appended the way [K] AppendSupervision does
  (``fst::Concat`` + ``RmEpsilon`` + breadth-first state order), which is the merged acceptor
  ``tc_supervision_create`` takes.

All format knowledge is restated from Kaldi's / OpenFst's published sources (kaldi ``base/io-funcs``,
``matrix/compressed-matrix.cc``, ``nnet3/nnet-example.cc``, ``nnet3/nnet-chain-example.cc``,
``chain/chain-supervision.cc``; OpenFst ``compact-fst.h``); it could not be checked against files written by
Kaldi in this environment (tests use ``tests/kaldi_egs_writer.py``, written from the same description).
The per-frame ``deriv_weights`` and the ``indexes`` the reference drops (``README.md:41``) are kept and returned.
"""
import io as _pyio
import os
import struct
import subprocess

import numpy as np

from .synth import SupFst


class EgsFormatError(ValueError):
    pass


# ---- Kaldi binary primitives (base/io-funcs-inl.h) -------------------------------------------------------
class _Reader:
    def __init__(self, stream):
        self.f = stream

    def read(self, n):
        b = self.f.read(n)
        if len(b) != n:
            raise EgsFormatError("unexpected end of stream")
        return b

    def peek(self, n=1):
        b = self.f.peek(n)[:n] if hasattr(self.f, "peek") else None
        if b is None:
            pos = self.f.tell()
            b = self.f.read(n)
            self.f.seek(pos)
        return b

    def token(self):
        out = bytearray()
        while True:
            c = self.f.read(1)
            if not c:
                raise EgsFormatError("unexpected end of stream in a token")
            if c in b" \t\n":
                if out:
                    return out.decode()
                continue
            out += c

    def expect(self, tok):
        t = self.token()
        if t != tok:
            raise EgsFormatError("expected %s, got %s" % (tok, t))

    def basic(self, fmt):
        size = struct.calcsize(fmt)
        n = self.read(1)[0]
        if n != size:
            raise EgsFormatError("basic type of size %d where %d was expected" % (n, size))
        return struct.unpack("<" + fmt, self.read(size))[0]

    def int32(self):
        return self.basic("i")

    def float32(self):
        return self.basic("f")

    def boolean(self):
        c = self.read(1)
        if c not in b"TF":
            raise EgsFormatError("bad bool")
        if self.peek(1) == b" ":
            self.read(1)
        return c == b"T"

    def array(self, dtype, count):
        dtype = np.dtype(dtype)
        return np.frombuffer(self.read(dtype.itemsize * int(count)), dtype=dtype, count=int(count)).copy()


def _read_index_vector(r):
    """[K] ReadIndexVector (nnet3/nnet-common.cc): ``<I1V>`` size, then per element one signed char (t relative to the
    previous index; the first one absolute with n = x = 0) or 127 followed by explicit (n, t, x)."""
    r.expect("<I1V>")
    size = r.int32()
    out = np.zeros((size, 3), np.int32)
    n = t = x = 0
    for i in range(size):
        c = struct.unpack("b", r.read(1))[0]
        if abs(c) < 125:
            if i == 0:
                n, t, x = 0, c, 0
            else:
                t = t + c
        else:
            if c != 127:
                raise EgsFormatError("bad index vector element")
            n, t, x = r.int32(), r.int32(), r.int32()
        out[i] = (n, t, x)
    return out


def _read_general_matrix(r):
    """[K] GeneralMatrix::Read: a full matrix (``FM`` float / ``DM`` double) or a CompressedMatrix (``CM``: one byte per
    element with per-column 4-point headers, ``CM2``: uint16, ``CM3``: uint8; matrix/compressed-matrix.cc)."""
    tok = r.token()
    if tok in ("FM", "DM"):
        rows, cols = r.int32(), r.int32()
        data = r.array(np.float32 if tok == "FM" else np.float64, rows * cols)
        return data.reshape(rows, cols).astype(np.float32)
    if tok in ("CM", "CM2", "CM3"):
        min_value, rng, rows, cols = struct.unpack("<ffii", r.read(16))
        if tok == "CM":
            hdr = r.array(np.uint16, 4 * cols).reshape(cols, 4).astype(np.float32)
            p = min_value + rng * hdr / 65535.0  # percentiles 0, 25, 75, 100 per column
            b = r.array(np.uint8, rows * cols).reshape(cols, rows).astype(np.float32)  # column-major bytes
            p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
            lo = p0 + (p25 - p0) * b * (1.0 / 64.0)
            mid = p25 + (p75 - p25) * (b - 64.0) * (1.0 / 128.0)
            hi = p75 + (p100 - p75) * (b - 192.0) * (1.0 / 63.0)
            return np.where(b <= 64, lo, np.where(b <= 192, mid, hi)).T.astype(np.float32).copy()
        if tok == "CM2":
            u = r.array(np.uint16, rows * cols).reshape(rows, cols).astype(np.float32)
            return (min_value + rng * u / 65535.0).astype(np.float32)
        u = r.array(np.uint8, rows * cols).reshape(rows, cols).astype(np.float32)
        return (min_value + rng * u / 255.0).astype(np.float32)
    raise EgsFormatError("unsupported matrix type %r (sparse features are not used by chain egs)" % tok)


def _read_compact_acceptor(r):
    """OpenFst ``CompactFst<StdArc, AcceptorCompactor>`` as [K] Supervision::Write stores the numerator FST: FstHeader,
    (numstates + 1) uint32 offsets, then {int32 label, float weight, int32 nextstate} elements; an element with
    label -1 carries a state's final weight.  Returns CSR arrays (arc_begin, ilabel, weight, nextstate, final)."""
    magic = struct.unpack("<i", r.read(4))[0]
    if magic != 2125659606:
        raise EgsFormatError("bad FST magic")

    def fst_string():
        n = struct.unpack("<i", r.read(4))[0]
        if n < 0 or n > 1 << 16:
            raise EgsFormatError("bad FST header string")
        return r.read(n).decode()

    fsttype, arctype = fst_string(), fst_string()
    version, flags = struct.unpack("<ii", r.read(8))
    _props, _start, nstates, _narcs = struct.unpack("<Qqqq", r.read(32))
    if fsttype != "compact_acceptor" or arctype != "standard" or (flags & 4) or _start not in (0, -1):
        raise EgsFormatError("supervision FST must be an unaligned compact_acceptor over StdArc starting at state 0 "
                             "(got %s/%s, start %d)" % (fsttype, arctype, _start))
    if flags & 1 or flags & 2:
        raise EgsFormatError("symbol tables inside a supervision FST are not supported")
    states = r.array(np.uint32, nstates + 1).astype(np.int64)
    ncomp = int(states[nstates]) if nstates > 0 else 0
    comp = np.frombuffer(r.read(12 * ncomp), dtype=np.dtype([("label", "<i4"), ("weight", "<f4"), ("next", "<i4")]))
    is_final = comp["label"] == -1
    final = np.full(nstates, np.inf, np.float32)
    owner = np.repeat(np.arange(nstates), np.diff(states))
    final[owner[is_final]] = comp["weight"][is_final]
    arcs = ~is_final
    arc_begin = np.zeros(nstates + 1, np.int32)
    np.add.at(arc_begin, owner[arcs] + 1, 1)
    arc_begin = np.cumsum(arc_begin).astype(np.int32)
    return (arc_begin, comp["label"][arcs].astype(np.int32), comp["weight"][arcs].astype(np.float32),
            comp["next"][arcs].astype(np.int32), final)


def _read_supervision(r):
    """[K] chain::Supervision::Read (chain/chain-supervision.cc), binary mode."""
    r.expect("<Supervision>")
    r.expect("<Weight>")
    weight = r.float32()
    r.expect("<NumSequences>")
    S = r.int32()
    r.expect("<FramesPerSeq>")
    T = r.int32()
    r.expect("<LabelDim>")
    P = r.int32()
    if r.peek(2) == b"<E":  # later Kaldi: <End2End> flag
        r.expect("<End2End>")
        if r.boolean():
            raise EgsFormatError("end-to-end (e2e) supervisions are outside this path (SURVEY.md section 8a caveat 3)")
    arc_begin, ilabel, w, nxt, final = _read_compact_acceptor(r)
    r.expect("</Supervision>")
    return SupFst(float(weight), S, T, P, len(final), arc_begin, ilabel, w, nxt, final)


def _read_vector_as_char(r):
    size_byte = r.read(1)[0]
    if size_byte != 1:
        raise EgsFormatError("bad <DW> vector")
    n = struct.unpack("<i", r.read(4))[0]
    return r.array(np.uint8, n).astype(np.float32) / 255.0


def read_chain_example(stream):
    """One binary ``NnetChainExample`` from ``stream`` (positioned after the ``\\0B`` marker).  Returns
    ``dict(inputs=[dict(name, indexes, features)], outputs=[dict(name, indexes, supervision, deriv_weights)])``."""
    r = stream if isinstance(stream, _Reader) else _Reader(stream)
    r.expect("<Nnet3ChainEg>")
    r.expect("<NumInputs>")
    inputs = []
    for _ in range(r.int32()):
        r.expect("<NnetIo>")
        name = r.token()
        idx = _read_index_vector(r)
        feats = _read_general_matrix(r)
        r.expect("</NnetIo>")
        inputs.append(dict(name=name, indexes=idx, features=feats))
    r.expect("<NumOutputs>")
    outputs = []
    for _ in range(r.int32()):
        r.expect("<NnetChainSup>")
        name = r.token()
        idx = _read_index_vector(r)
        sup = _read_supervision(r)
        tok = r.token()
        if tok == "<DW>":
            dw = _read_vector_as_char(r)
            r.expect("</NnetChainSup>")
        elif tok == "<DW2>":
            r.expect("FV")
            dw = r.array(np.float32, r.int32())
            r.expect("</NnetChainSup>")
        elif tok == "</NnetChainSup>":
            dw = np.ones(len(idx), np.float32)
        else:
            raise EgsFormatError("unexpected token %s in <NnetChainSup>" % tok)
        outputs.append(dict(name=name, indexes=idx, supervision=sup, deriv_weights=dw))
    r.expect("</Nnet3ChainEg>")
    return dict(inputs=inputs, outputs=outputs)


# ---- tables -----------------------------------------------------------------------------------------------
def _split_rspecifier(rspec):
    """'ark:foo', 'ark,bg:foo', 'scp:foo', or a bare archive path -> (kind, path)."""
    if ":" in rspec and rspec.split(":", 1)[0].split(",")[0] in ("ark", "scp"):
        head, path = rspec.split(":", 1)
        return head.split(",")[0], path
    return "ark", rspec


def _open_rx(path):
    """Kaldi rxfilename: a file, or 'command |'."""
    path = path.strip()
    if path.endswith("|"):
        proc = subprocess.Popen(path[:-1], shell=True, stdout=subprocess.PIPE)
        return _pyio.BufferedReader(proc.stdout), proc
    return open(path, "rb"), None


def _expect_binary(r):
    if r.read(2) != b"\0B":
        raise EgsFormatError("text-mode egs are not supported (expected the \\0B binary marker)")


def iter_archive(path):
    """(key, example) pairs of a sequential binary archive."""
    f, proc = _open_rx(path)
    try:
        r = _Reader(f)
        while True:
            c = f.read(1)
            while c in (b" ", b"\n"):
                c = f.read(1)
            if not c:
                return
            key = bytearray(c)
            while True:
                c = f.read(1)
                if not c:
                    raise EgsFormatError("archive ends inside a key")
                if c == b" ":
                    break
                key += c
            _expect_binary(r)
            yield key.decode(), read_chain_example(r)
    finally:
        f.close()
        if proc is not None:
            proc.wait()


def read_scp(path):
    """``key path:offset`` (or ``key path``) lines -> ordered list of (key, path, offset)."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            key, loc = line.split(None, 1)
            if ":" in loc and loc.rsplit(":", 1)[1].isdigit():
                p, off = loc.rsplit(":", 1)
                out.append((key, p, int(off)))
            else:
                out.append((key, loc, None))
    return out


def read_scp_entry(path, offset):
    with open(path, "rb") as f:
        if offset is not None:
            f.seek(offset)
        r = _Reader(f)
        _expect_binary(r)
        return read_chain_example(r)


def iter_rspecifier(rspec):
    kind, path = _split_rspecifier(rspec)
    if kind == "ark":
        for kv in iter_archive(path):
            yield kv
    else:
        for key, p, off in read_scp(path):
            yield key, read_scp_entry(p, off)


# ---- merging ([K] MergeChainExamples / AppendSupervision) ----------------------------------------------------
def _fst_state_times(sup):
    times = np.full(sup.num_states, -1, np.int64)
    times[0] = 0
    for s in range(sup.num_states):
        if times[s] < 0:
            raise EgsFormatError("supervision FST is not connected / not time-sorted")
        a0, a1 = sup.arc_begin[s], sup.arc_begin[s + 1]
        nt = times[sup.nextstate[a0:a1]]
        if np.any((nt >= 0) & (nt != times[s] + 1)):
            raise EgsFormatError("supervision FST paths have unequal lengths")
        times[sup.nextstate[a0:a1]] = times[s] + 1
    return times


def append_supervisions(sups):
    """[K] AppendSupervision for examples of equal weight, frames-per-sequence and label-dim: the FSTs are
    concatenated (``fst::Concat``), epsilons removed -- every final state f of piece k-1 (final weight w_f) receives
    copies of piece k's start arcs with weight w_f + arc weight and stops being final; piece k's start state
    disappears -- and states are renumbered breadth-first, i.e. in time order.  Returns one merged ``SupFst``."""
    if not sups:
        raise ValueError("nothing to merge")
    w, T, P = sups[0].weight, sups[0].frames_per_sequence, sups[0].label_dim
    for s in sups:
        if (s.weight, s.frames_per_sequence, s.label_dim) != (w, T, P):
            raise EgsFormatError("cannot merge supervisions with different weight / frames / label-dim")
    if len(sups) == 1:
        return sups[0]
    # global ids: piece 0 keeps all its states; later pieces drop their start state (id 0)
    pieces = []
    base = 0
    for k, s in enumerate(sups):
        times = _fst_state_times(s)
        total = s.num_sequences * T
        finals = np.flatnonzero(~np.isinf(s.final))
        if np.any(times[finals] != total) or np.any(np.diff(s.arc_begin)[finals] != 0):
            raise EgsFormatError("final states of a supervision must sit at the last frame and have no arcs")
        if k > 0 and np.any(s.nextstate == 0):
            raise EgsFormatError("supervision start state has incoming arcs")
        gid = np.arange(s.num_states, dtype=np.int64) + base - (1 if k > 0 else 0)
        pieces.append((s, times, finals, gid))
        base += s.num_states - (1 if k > 0 else 0)
    n_total = base
    out_il, out_w, out_next, counts = [], [], [], np.zeros(n_total, np.int64)
    final = np.full(n_total, np.inf, np.float32)
    order_time = np.zeros(n_total, np.int64)
    rows = [[] for _ in range(n_total)]
    toff = 0
    for k, (s, times, finals, gid) in enumerate(pieces):
        for st in range(1 if k > 0 else 0, s.num_states):
            g = int(gid[st])
            order_time[g] = toff + times[st]
            for a in range(s.arc_begin[st], s.arc_begin[st + 1]):
                rows[g].append((int(s.ilabel[a]), float(s.arc_weight[a]), int(gid[s.nextstate[a]])))
        if k + 1 < len(pieces):
            nxt, _, _, ngid = pieces[k + 1]
            for f in finals:
                g = int(gid[f])
                for a in range(nxt.arc_begin[0], nxt.arc_begin[1]):
                    rows[g].append((int(nxt.ilabel[a]), float(s.final[f]) + float(nxt.arc_weight[a]),
                                    int(ngid[nxt.nextstate[a]])))
        else:
            final[gid[finals]] = s.final[finals]
        toff += s.num_sequences * T
    # breadth-first (time) order, stable inside a time
    perm = np.argsort(order_time, kind="stable")
    newid = np.empty(n_total, np.int64)
    newid[perm] = np.arange(n_total)
    arc_begin = [0]
    for g in perm:
        for (il, aw, nx) in rows[g]:
            out_il.append(il)
            out_w.append(aw)
            out_next.append(int(newid[nx]))
        arc_begin.append(len(out_il))
    return SupFst(float(w), sum(s.num_sequences for s in sups), T, P, n_total, np.asarray(arc_begin, np.int32),
                  np.asarray(out_il, np.int32), np.asarray(out_w, np.float32), np.asarray(out_next, np.int32),
                  final[perm])


def merge_chain_examples(examples):
    """[K] MergeChainExamples for examples with one output: inputs with the same name are stacked example by example
    (row blocks; the reference then views them as (batch, time, feat), ``io.py:98-103``); the ``n`` of every index is the
    example's position in the batch; supervisions are appended; ``deriv_weights`` and output indexes are re-ordered
    frame-major ((frame 0 of every sequence), (frame 1 ...), as in [K] NnetChainSupervision) ."""
    if not examples:
        raise ValueError("nothing to merge")
    names = [io_["name"] for io_ in examples[0]["inputs"]]
    inputs = []
    for j, name in enumerate(names):
        feats, idx = [], []
        for n, eg in enumerate(examples):
            io_ = eg["inputs"][j]
            if io_["name"] != name:
                raise EgsFormatError("examples disagree on input names")
            feats.append(io_["features"])
            ix = io_["indexes"].copy()
            ix[:, 0] = n
            idx.append(ix)
        inputs.append(dict(name=name, indexes=np.concatenate(idx), features=np.concatenate(feats)))
    outs = [eg["outputs"][0] for eg in examples]
    sup = append_supervisions([o["supervision"] for o in outs])
    T = sup.frames_per_sequence
    idx_seq, dw_seq = [], []
    n = 0
    for o in outs:
        S_o = o["supervision"].num_sequences
        ix = o["indexes"].reshape(T, S_o, 3).copy()  # frame-major inside the example
        ix[:, :, 0] = np.arange(n, n + S_o)[None, :]
        idx_seq.append(ix)
        dw_seq.append(o["deriv_weights"].reshape(T, S_o))
        n += S_o
    indexes = np.concatenate(idx_seq, axis=1).reshape(-1, 3)
    deriv_weights = np.concatenate(dw_seq, axis=1).reshape(-1)
    return dict(inputs=inputs, outputs=[dict(name=outs[0]["name"], indexes=indexes, supervision=sup,
                                             deriv_weights=deriv_weights)])
