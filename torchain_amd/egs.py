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
        """The next ``n`` bytes without consuming them.  ``BufferedReader.peek`` returns only what is left in its
        buffer -- possibly ONE byte at a block boundary -- so a short answer is completed by read + seek where the
        stream can seek; callers on pipes must not ask for more than one byte."""
        b = self.f.peek(n)[:n] if hasattr(self.f, "peek") else b""
        if len(b) < n and (not hasattr(self.f, "seekable") or self.f.seekable()):
            pos = self.f.tell()
            b = self.f.read(n)
            self.f.seek(pos)
        return b

    def peek_some(self, n):
        """Between 1 and ``n`` of the next bytes without consuming them (whatever the buffer holds; at least one)."""
        b = self.f.peek(n)[:n] if hasattr(self.f, "peek") else b""
        return b if b else self.peek(1) or self.read(1)[:0]

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
    n = t = x = 0  # (the first element is "absolute" relative to n = t = x = 0)
    i = 0
    while i < size:
        # a run of one-byte elements (the usual case: all of them but the first of each sequence), decoded at once
        chunk = r.peek_some(size - i)
        c = np.frombuffer(chunk, np.int8).astype(np.int32)
        stop = np.flatnonzero(np.abs(c) >= 125)
        run = int(stop[0]) if len(stop) else len(c)
        if run > 0:
            ts = t + np.cumsum(c[:run])
            out[i:i + run, 0] = n
            out[i:i + run, 1] = ts
            out[i:i + run, 2] = x
            t = int(ts[-1])
            r.read(run)
            i += run
            continue
        if r.read(1) != b"\x7f":
            raise EgsFormatError("bad index vector element")
        n, t, x = r.int32(), r.int32(), r.int32()
        out[i] = (n, t, x)
        i += 1
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
    # later Kaldi: <End2End> flag.  One byte decides: the FST that follows otherwise starts with its magic number
    # 2125659606 = d6 fd b2 7e, never '<' (and a buffered reader's peek may hold a single byte at a block boundary).
    if r.peek(1) == b"<":
        r.expect("<End2End>")
        if r.boolean():
            raise EgsFormatError("end-to-end (e2e) supervisions are outside this path (SURVEY.md section 8a caveat 3)")
    arc_begin, ilabel, w, nxt, final = _read_compact_acceptor(r)
    tok = r.token()
    if tok == "<AlignmentPdfs>":  # later Kaldi, behind the FST when not empty ([K] WriteIntegerVector); unused here
        if r.read(1)[0] != 4:
            raise EgsFormatError("<AlignmentPdfs>: element size")
        n = struct.unpack("<i", r.read(4))[0]
        if n < 0 or n > S * T:
            raise EgsFormatError("<AlignmentPdfs>: size")
        r.read(4 * n)
        tok = r.token()
    if tok != "</Supervision>":
        raise EgsFormatError("expected </Supervision>, got %s" % tok)
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
        for kv in iter_archive_native(path):  # (iter_archive is the numpy statement of the same format)
            yield kv
    else:
        for key, p, off in read_scp(path):
            yield key, read_merged_native([(p, off)], merge_single=False)  # (read_scp_entry: the numpy statement)


# ---- merging ([K] MergeChainExamples / AppendSupervision) ----------------------------------------------------
def _state_times(n_states, src, dst, start_ids, max_time):
    """Time (distance from its piece's start state) of every state of a set of time-sorted acceptors held in common
    arrays: one predecessor per state (any: all paths into a state are equally long), then pointer doubling --
    log2(max_time) numpy passes instead of a Python loop over states.  ``src`` / ``dst`` are the arcs' end points."""
    pred = np.full(n_states, n_states, np.int32)
    pred[dst] = src  # (whichever of a state's predecessors is written last: all are equally far from the start)
    is_start = np.zeros(n_states, bool)
    is_start[start_ids] = True
    if np.any(is_start & (pred < n_states)):
        raise EgsFormatError("supervision start state has incoming arcs")
    if np.any(~is_start & (pred == n_states)):
        raise EgsFormatError("supervision FST is not connected / not time-sorted")
    anc = np.where(is_start, np.arange(n_states, dtype=np.int32), pred)   # roots point at themselves
    depth = (~is_start).astype(np.int32)
    for _ in range(max(1, int(max_time)).bit_length()):  # after k rounds a state has jumped 2^k levels
        depth += depth[anc]
        anc = anc[anc]
    if not np.all(is_start[anc]) or np.any(depth[dst] != depth[src] + 1):
        raise EgsFormatError("supervision FST paths have unequal lengths")
    return depth.astype(np.int64)


def _fst_state_times(sup):
    src = np.repeat(np.arange(sup.num_states, dtype=np.int64), np.diff(sup.arc_begin))
    return _state_times(sup.num_states, src, sup.nextstate.astype(np.int64), np.array([0]), sup.num_states)


def append_supervisions_numpy(sups):
    """[K] AppendSupervision for examples of equal weight, frames-per-sequence and label-dim: the FSTs are
    concatenated (``fst::Concat``), epsilons removed -- every final state f of piece k-1 (final weight w_f) receives
    copies of piece k's start arcs with weight w_f + arc weight and stops being final; piece k's start state
    disappears -- and states are renumbered breadth-first, i.e. in time order.  Returns one merged ``SupFst``.

    The numpy statement of ``append_supervisions`` (which runs natively in the library): all pieces live in common
    arrays, the only Python loop runs over the pieces.  Kept as the independent check of the native merge in the tests."""
    if not sups:
        raise ValueError("nothing to merge")
    w, T, P = sups[0].weight, sups[0].frames_per_sequence, sups[0].label_dim
    for s in sups:
        if (s.weight, s.frames_per_sequence, s.label_dim) != (w, T, P):
            raise EgsFormatError("cannot merge supervisions with different weight / frames / label-dim")
    if len(sups) == 1:
        return sups[0]
    K = len(sups)
    nst = np.array([s.num_states for s in sups], np.int64)
    narc = np.array([len(s.ilabel) for s in sups], np.int64)
    soff = np.concatenate([[0], np.cumsum(nst)])            # raw global id of piece k's state 0
    aoff = np.concatenate([[0], np.cumsum(narc)])
    n_raw = int(soff[-1])
    piece_of_state = np.repeat(np.arange(K), nst)
    deg = np.concatenate([np.diff(s.arc_begin) for s in sups]).astype(np.int64)
    src = np.repeat(np.arange(n_raw, dtype=np.int64), deg)
    piece_of_arc = np.repeat(np.arange(K), narc)
    dst = np.concatenate([s.nextstate for s in sups]).astype(np.int64) + soff[piece_of_arc]
    il = np.concatenate([s.ilabel for s in sups]).astype(np.int32)
    aw = np.concatenate([s.arc_weight for s in sups]).astype(np.float32)
    fin = np.concatenate([s.final for s in sups]).astype(np.float32)
    total = np.array([s.num_sequences * T for s in sups], np.int64)
    times = _state_times(n_raw, src, dst, soff[:-1], int(total.max()) + 1)
    is_final = ~np.isinf(fin)
    if np.any(times[is_final] != total[piece_of_state[is_final]]) or np.any(deg[is_final] != 0):
        raise EgsFormatError("final states of a supervision must sit at the last frame and have no arcs")
    toff = np.concatenate([[0], np.cumsum(total)])[:-1]
    gtime = times + toff[piece_of_state]
    # states that survive: everything but the start states of pieces 1 .. K-1; new ids in (time, raw id) order
    keep = np.ones(n_raw, bool)
    keep[soff[1:-1]] = False
    kept = np.flatnonzero(keep)
    order = kept[np.argsort(gtime[kept], kind="stable")]
    newid = np.full(n_raw, -1, np.int64)
    newid[order] = np.arange(len(order))
    n_total = len(order)
    # arcs that stay: all but the start arcs of pieces 1 .. K-1
    stay = keep[src]
    m_src, m_il, m_w, m_dst = [newid[src[stay]]], [il[stay]], [aw[stay]], [newid[dst[stay]]]
    # bridges: every final state of piece k-1 gets piece k's start arcs
    for k in range(1, K):
        f = np.flatnonzero(is_final[soff[k - 1]:soff[k]]) + soff[k - 1]
        a0, a1 = aoff[k], aoff[k] + deg[soff[k]]
        na = int(a1 - a0)
        if len(f) == 0 or na == 0:
            continue
        nf = len(f)
        m_src.append(np.broadcast_to(newid[f][:, None], (nf, na)).reshape(-1))
        m_il.append(np.broadcast_to(il[a0:a1][None, :], (nf, na)).reshape(-1))
        m_w.append((fin[f][:, None] + aw[a0:a1][None, :]).astype(np.float32).reshape(-1))
        m_dst.append(np.broadcast_to(newid[dst[a0:a1]][None, :], (nf, na)).reshape(-1))
    m_src = np.concatenate(m_src)
    perm = np.argsort(m_src, kind="stable")  # by state; inside a state the order the pieces gave
    arc_begin = np.zeros(n_total + 1, np.int64)
    np.cumsum(np.bincount(m_src, minlength=n_total), out=arc_begin[1:])
    final = np.full(n_total, np.inf, np.float32)
    last = np.flatnonzero(is_final[soff[K - 1]:]) + soff[K - 1]
    final[newid[last]] = fin[last]
    return SupFst(float(w), int(sum(s.num_sequences for s in sups)), T, P, n_total, arc_begin.astype(np.int32),
                  np.concatenate(m_il)[perm].astype(np.int32), np.concatenate(m_w)[perm].astype(np.float32),
                  np.concatenate(m_dst)[perm].astype(np.int32), final)


def append_supervisions(sups):
    """[K] AppendSupervision (see ``append_supervisions_numpy``) through the library's ``tc_supervision_append``: the
    reference merges natively too (``src/my_lib_example_rand.cpp:160``).  Host memory only."""
    import ctypes as C

    from ._lib import lib

    if not sups:
        raise ValueError("nothing to merge")
    w, T, P = sups[0].weight, sups[0].frames_per_sequence, sups[0].label_dim
    for s in sups:
        if (s.weight, s.frames_per_sequence, s.label_dim) != (w, T, P):
            raise EgsFormatError("cannot merge supervisions with different weight / frames / label-dim")
    if len(sups) == 1:
        return sups[0]
    K = len(sups)
    nst = np.array([s.num_states for s in sups], np.int32)
    frames = np.array([s.num_sequences * T for s in sups], np.int32)
    ab = np.ascontiguousarray(np.concatenate([s.arc_begin for s in sups]), np.int32)
    il = np.ascontiguousarray(np.concatenate([s.ilabel for s in sups]), np.int32)
    aw = np.ascontiguousarray(np.concatenate([s.arc_weight for s in sups]), np.float32)
    nx = np.ascontiguousarray(np.concatenate([s.nextstate for s in sups]), np.int32)
    fin = np.ascontiguousarray(np.concatenate([s.final for s in sups]), np.float32)
    # bound of the merged arcs: all arcs + (final states before a boundary) x (start arcs behind it)
    n_fin = np.array([int(np.count_nonzero(~np.isinf(s.final))) for s in sups[:-1]], np.int64)
    n_start = np.array([int(s.arc_begin[1]) if s.num_states > 0 else 0 for s in sups[1:]], np.int64)
    cap_states, cap_arcs = int(nst.sum()), int(len(il) + int((n_fin * n_start).sum()))
    o_ab = np.empty(cap_states + 1, np.int32)
    o_il, o_w, o_nx = np.empty(cap_arcs, np.int32), np.empty(cap_arcs, np.float32), np.empty(cap_arcs, np.int32)
    o_fin = np.empty(cap_states, np.float32)
    n_states, n_arcs = C.c_int32(0), C.c_int64(0)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = lib.tc_supervision_append(K, p(nst), p(frames), p(ab), p(il), p(aw), p(nx), p(fin), cap_states, cap_arcs,
                                   C.cast(C.byref(n_states), C.c_void_p), C.cast(C.byref(n_arcs), C.c_void_p),
                                   p(o_ab), p(o_il), p(o_w), p(o_nx), p(o_fin))
    if rc == -2:
        raise EgsFormatError("a supervision FST is not a connected acceptor whose paths all have num_sequences * "
                             "frames_per_sequence arcs, with arc-less final states at the last frame")
    if rc != 0:
        raise EgsFormatError("tc_supervision_append failed: %d" % rc)
    ns, na = n_states.value, n_arcs.value
    return SupFst(float(w), int(sum(s.num_sequences for s in sups)), T, P, ns, o_ab[:ns + 1].copy(), o_il[:na].copy(),
                  o_w[:na].copy(), o_nx[:na].copy(), o_fin[:ns].copy())


class _NativeExample:
    """Owns one ``tc_example`` (freed with the last array that views its memory)."""

    def __init__(self, handle, free):
        self._handle, self._free = handle, free

    def __del__(self):
        if self._handle:
            self._free(self._handle)
            self._handle = None


def read_merged_native(entries, merge_single=True):
    """The minibatch of the scp entries ``[(path, offset), ...]`` read and merged by the library's native reader
    (``tc_example_read``: what the reference does through Kaldi in ``src/my_lib_example_rand.cpp:35-177``) -- the same
    dict ``merge_chain_examples([read_scp_entry(p, o) for p, o in entries])`` returns.  Runs without the interpreter
    lock apart from the final copies into numpy arrays."""
    import ctypes as C

    from ._lib import lib

    n = len(entries)
    if n == 0:
        raise ValueError("nothing to read")
    paths = (C.c_char_p * n)(*[os.fsencode(p) for p, _ in entries])
    offsets = (C.c_int64 * n)(*[-1 if o is None else int(o) for _, o in entries])
    handle = C.c_void_p()
    rc = lib.tc_example_read(C.cast(paths, C.c_void_p), C.cast(offsets, C.c_void_p), n, int(bool(merge_single)),
                             C.cast(C.byref(handle), C.c_void_p))
    if rc != 0:
        msg = (lib.tc_example_last_error() or b"").decode()
        if rc == -6:
            raise OSError(msg or "cannot read an example")
        raise EgsFormatError(msg or "tc_example_read failed: %d" % rc)
    return _wrap_native(handle)


def _wrap_native(handle):
    """A ``tc_example`` as the dict the numpy reader returns; takes ownership of the handle."""
    import ctypes as C

    from ._lib import lib

    owner = _NativeExample(handle, lib.tc_example_free)
    try:
        def arr(ptr, ctype, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).astype(dtype, copy=True)

        def view(ptr, ctype, count, dtype):
            # the large arrays (stacked features: megabytes per minibatch) are not copied: the array's buffer keeps the
            # native object alive for as long as the array (or anything sliced from it) lives
            if count == 0:
                return np.zeros(0, dtype)
            buf = (ctype * count).from_address(ptr.value)
            buf._owner = owner
            return np.frombuffer(buf, dtype=dtype)

        counts = (C.c_int32 * 2)()
        lib.tc_example_counts(handle, C.cast(counts, C.c_void_p))
        ref = lambda x: C.cast(C.byref(x), C.c_void_p)
        inputs, outputs = [], []
        for j in range(counts[0]):
            name, feats, idx = C.c_char_p(), C.c_void_p(), C.c_void_p()
            rows, cols, nidx = C.c_int32(), C.c_int32(), C.c_int32()
            lib.tc_example_input(handle, j, ref(name), ref(rows), ref(cols), ref(nidx), ref(feats), ref(idx))
            inputs.append(dict(name=name.value.decode(),
                               indexes=arr(idx, C.c_int32, 3 * nidx.value, np.int32).reshape(-1, 3),
                               features=view(feats, C.c_float, rows.value * cols.value, np.float32).reshape(rows.value, cols.value)))
        for j in range(counts[1]):
            name, idx, dw = C.c_char_p(), C.c_void_p(), C.c_void_p()
            ab, il, aw, nx, fin = (C.c_void_p() for _ in range(5))
            nidx, weight, dims = C.c_int32(), C.c_float(), (C.c_int32 * 5)()
            lib.tc_example_output(handle, j, ref(name), ref(nidx), ref(idx), ref(dw), ref(weight), C.cast(dims, C.c_void_p),
                                  ref(ab), ref(il), ref(aw), ref(nx), ref(fin))
            S, T, P, ns, na = (int(v) for v in dims)
            sup = SupFst(float(weight.value), S, T, P, ns, arr(ab, C.c_int32, ns + 1, np.int32),
                         arr(il, C.c_int32, na, np.int32), arr(aw, C.c_float, na, np.float32),
                         arr(nx, C.c_int32, na, np.int32), arr(fin, C.c_float, ns, np.float32))
            outputs.append(dict(name=name.value.decode(), indexes=arr(idx, C.c_int32, 3 * nidx.value, np.int32).reshape(-1, 3),
                                supervision=sup, deriv_weights=arr(dw, C.c_float, nidx.value, np.float32)))
        return dict(inputs=inputs, outputs=outputs)
    finally:
        del owner  # (freed now unless a feature array holds it)


def iter_archive_native(path):
    """(key, example) pairs of a sequential binary archive or of a ``command |`` through the library's reader
    (``tc_archive_*``: the reference reads these through Kaldi's SequentialNnetChainExampleReader,
    ``src/my_lib_example.cpp:35-69``)."""
    import ctypes as C

    from ._lib import lib

    arch = C.c_void_p()
    rc = lib.tc_archive_open(os.fsencode(path), C.cast(C.byref(arch), C.c_void_p))
    if rc != 0:
        raise OSError((lib.tc_example_last_error() or b"cannot open archive").decode())
    key = C.create_string_buffer(1 << 16)
    try:
        while True:
            handle = C.c_void_p()
            rc = lib.tc_archive_next(arch, C.cast(key, C.c_void_p), 1 << 16, C.cast(C.byref(handle), C.c_void_p))
            if rc == 0:
                return
            if rc < 0:
                raise EgsFormatError((lib.tc_example_last_error() or b"").decode() or "tc_archive_next failed: %d" % rc)
            yield key.value.decode(), _wrap_native(handle)
    finally:
        lib.tc_archive_close(arch)


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
