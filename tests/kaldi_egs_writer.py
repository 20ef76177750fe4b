"""Writer of Kaldi chain egs for the tests (the counterpart of torchain_amd/egs.py, restated from the same published
format descriptions: kaldi base/io-funcs, matrix/compressed-matrix.cc, nnet3/nnet-chain-example.cc,
chain/chain-supervision.cc, OpenFst compact-fst.h).  Test infrastructure only."""
import struct

import numpy as np


def _tok(t):
    return t.encode() + b" "


def _basic(fmt, v):
    return bytes([struct.calcsize(fmt)]) + struct.pack("<" + fmt, v)


def index_vector(idx):
    out = _tok("<I1V>") + _basic("i", len(idx))
    pn = pt = px = 0
    for i, (n, t, x) in enumerate(idx):
        n, t, x = int(n), int(t), int(x)
        if i == 0:
            ok = n == 0 and x == 0 and abs(t) < 125
            delta = t
        else:
            ok = n == pn and x == px and abs(t - pt) < 125
            delta = t - pt
        out += struct.pack("b", delta) if ok else struct.pack("b", 127) + _basic("i", n) + _basic("i", t) + _basic("i", x)
        pn, pt, px = n, t, x
    return out


def general_matrix(m, kind="FM"):
    m = np.asarray(m, np.float32)
    rows, cols = m.shape
    if kind == "FM":
        return _tok("FM") + _basic("i", rows) + _basic("i", cols) + m.tobytes()
    if kind == "DM":
        return _tok("DM") + _basic("i", rows) + _basic("i", cols) + m.astype(np.float64).tobytes()
    mn, mx = float(m.min()), float(m.max())
    rng = mx - mn if mx > mn else 1.0
    hdr = struct.pack("<ffii", mn, rng, rows, cols)
    if kind == "CM2":
        u = np.clip(np.rint((m - mn) / rng * 65535.0), 0, 65535).astype(np.uint16)
        return _tok("CM2") + hdr + u.tobytes()
    if kind == "CM3":
        u = np.clip(np.rint((m - mn) / rng * 255.0), 0, 255).astype(np.uint8)
        return _tok("CM3") + hdr + u.tobytes()
    assert kind == "CM"
    q = np.sort(m, axis=0)
    pct = np.stack([q[0], q[rows // 4], q[(3 * rows) // 4], q[rows - 1]], axis=1)  # (cols, 4)
    u16 = np.clip(np.rint((pct - mn) / rng * 65535.0), 0, 65535).astype(np.int64)
    for k in range(1, 4):  # strictly increasing, as [K] ComputeColHeader enforces
        u16[:, k] = np.maximum(u16[:, k], u16[:, k - 1] + 1)
    u16 = np.minimum(u16, [65532, 65533, 65534, 65535])
    p = mn + rng * u16.astype(np.float32) / 65535.0
    p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
    v = m.T  # (cols, rows)
    lo = np.rint((v - p0) / (p25 - p0) * 64.0)
    mid = 64 + np.rint((v - p25) / (p75 - p25) * 128.0)
    hi = 192 + np.rint((v - p75) / (p100 - p75) * 63.0)
    b = np.where(v < p25, np.clip(lo, 0, 64), np.where(v < p75, np.clip(mid, 64, 192), np.clip(hi, 192, 255)))
    return _tok("CM") + hdr + u16.astype(np.uint16).tobytes() + b.astype(np.uint8).tobytes()


def compact_acceptor(sup):
    """OpenFst CompactFst<StdArc, AcceptorCompactor, uint32>: header, state offsets, {label, weight, nextstate}."""
    def s(b):
        return struct.pack("<i", len(b)) + b
    elems = []
    offs = [0]
    for st in range(sup.num_states):
        if not np.isinf(sup.final[st]):
            elems.append((-1, float(sup.final[st]), -1))
        for a in range(sup.arc_begin[st], sup.arc_begin[st + 1]):
            elems.append((int(sup.ilabel[a]), float(sup.arc_weight[a]), int(sup.nextstate[a])))
        offs.append(len(elems))
    out = struct.pack("<i", 2125659606) + s(b"compact_acceptor") + s(b"standard")
    out += struct.pack("<iiQqqq", 2, 0, 0x0000000000010000, 0, sup.num_states, int(sup.arc_begin[-1]))
    out += np.asarray(offs, np.uint32).tobytes()
    for (l, w, n) in elems:
        out += struct.pack("<ifi", l, w, n)
    return out


def supervision(sup, e2e_flag=False, alignment_pdfs=None):
    out = _tok("<Supervision>") + _tok("<Weight>") + _basic("f", sup.weight)
    out += _tok("<NumSequences>") + _basic("i", sup.num_sequences) + _tok("<FramesPerSeq>") + _basic("i", sup.frames_per_sequence)
    out += _tok("<LabelDim>") + _basic("i", sup.label_dim)
    if e2e_flag:
        out += _tok("<End2End>") + b"F "
    out += compact_acceptor(sup)
    if alignment_pdfs is not None:  # later Kaldi: WriteToken("<AlignmentPdfs>"), WriteIntegerVector (binary)
        a = np.asarray(alignment_pdfs, np.int32)
        out += _tok("<AlignmentPdfs>") + bytes([4]) + struct.pack("<i", len(a)) + a.tobytes()
    return out + _tok("</Supervision>")


def chain_example(eg, matrix_kind="FM", dw="DW2", e2e_flag=False, alignment_pdfs=False):
    out = _tok("<Nnet3ChainEg>") + _tok("<NumInputs>") + _basic("i", len(eg["inputs"]))
    for io_ in eg["inputs"]:
        out += _tok("<NnetIo>") + _tok(io_["name"]) + index_vector(io_["indexes"]) + general_matrix(io_["features"], matrix_kind)
        out += _tok("</NnetIo>")
    out += _tok("<NumOutputs>") + _basic("i", len(eg["outputs"]))
    for o in eg["outputs"]:
        out += _tok("<NnetChainSup>") + _tok(o["name"]) + index_vector(o["indexes"])
        sup = o["supervision"]
        out += supervision(sup, e2e_flag, np.arange(sup.num_sequences * sup.frames_per_sequence) % sup.label_dim if alignment_pdfs else None)
        if dw == "DW2":
            out += _tok("<DW2>") + _tok("FV") + _basic("i", len(o["deriv_weights"])) + np.asarray(o["deriv_weights"], np.float32).tobytes()
        elif dw == "DW":
            b = np.clip(np.rint(np.asarray(o["deriv_weights"]) * 255.0), 0, 255).astype(np.uint8)
            out += _tok("<DW>") + bytes([1]) + struct.pack("<i", len(b)) + b.tobytes()
        out += _tok("</NnetChainSup>")
    return out + _tok("</Nnet3ChainEg>")


def write_ark(path, keyed_examples, scp_path=None, **kw):
    """Binary archive ``key SPACE \\0B object ...``; optionally the matching scp (``key path:offset``)."""
    lines = []
    with open(path, "wb") as f:
        for key, eg in keyed_examples:
            f.write(key.encode() + b" ")
            lines.append("%s %s:%d\n" % (key, path, f.tell()))
            f.write(b"\0B" + chain_example(eg, **kw))
    if scp_path:
        with open(scp_path, "w") as f:
            f.writelines(lines)
