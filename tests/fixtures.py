"""Fixtures shared by test modules (CPU-safe: numpy, the synthetic generators and the test-side Kaldi writer only): the committed
golden vectors, an OpenFst ``VectorFst<StdArc>`` writer, synthetic chain egs and archives of them.  Moved here in round 6 so that no
test module imports another."""
import glob
import os
import struct

import numpy as np

from torchain_amd import synth

import kaldi_egs_writer as kw

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_golden(path):
    z = np.load(path)
    fst = synth.DenFst(int(z["den_num_states"]), z["den_src"], z["den_dst"], z["den_ilabel"], z["den_weight"],
                       z["den_final"], int(z["den_start"]), int(z["num_pdfs"]))
    sup = synth.SupFst(float(z["sup_weight"]), int(z["num_sequences"]), int(z["frames_per_sequence"]),
                       int(z["num_pdfs"]), int(z["sup_num_states"]), z["sup_arc_begin"], z["sup_ilabel"],
                       z["sup_arc_weight"], z["sup_nextstate"], z["sup_final"])
    return z, fst, sup


def write_openfst_vector(path, fst, with_symbols=False, fst_type=b"vector", arc_type=b"standard", version=2,
                         truncate_to=None):
    """Writes an OpenFst binary VectorFst<StdArc> (what fst::ReadFstKaldi reads for a plain file): FstHeader
    {int32 magic, string fsttype, string arctype, int32 version, int32 flags, uint64 properties, int64 start,
    int64 numstates, int64 numarcs}, optional input / output symbol tables (flags bits 0 / 1), then per state
    {float final, int64 narcs, narcs x {int32 ilabel, int32 olabel, float weight, int32 nextstate}}."""
    def s(b):
        return struct.pack("<i", len(b)) + b

    def symtab(name, n):
        out = struct.pack("<i", 2125658996) + s(name) + struct.pack("<qq", n, n)
        for k in range(n):
            out += s(b"pdf%d" % k if k else b"<eps>") + struct.pack("<q", k)
        return out

    blob = struct.pack("<i", 2125659606) + s(fst_type) + s(arc_type)
    blob += struct.pack("<iiQqqq", version, 3 if with_symbols else 0, 0, int(fst.start), fst.num_states, len(fst.src))
    if with_symbols:
        blob += symtab(b"isyms", fst.num_pdfs + 1) + symtab(b"osyms", fst.num_pdfs + 1)
    first = np.searchsorted(fst.src, np.arange(fst.num_states + 1))
    for st in range(fst.num_states):
        blob += struct.pack("<fq", float(fst.final[st]), int(first[st + 1] - first[st]))
        for a in range(first[st], first[st + 1]):
            blob += struct.pack("<iifi", int(fst.ilabel[a]), int(fst.ilabel[a]), float(fst.weight[a]), int(fst.dst[a]))
    with open(path, "wb") as f:
        f.write(blob if truncate_to is None else blob[:truncate_to])
    return len(blob)


def make_example(fst, T, seed, n_seq=1, feat_dim=7, ivec_dim=3, left=4, weight=1.0, final_weights=False):
    """A synthetic chain eg: T output frames at t = 0, 3, 6, ... (frame-subsampling 3 as in the recipe), an input
    window of 3*T + 2*left frames, a one-row i-vector, a supervision of `n_seq` sequences."""
    rng = np.random.default_rng(seed)
    sup = synth.random_supervision(fst, n_seq, T, 2, seed=seed, weight=weight, final_weights=final_weights)
    n_in = 3 * T + 2 * left
    feats = rng.standard_normal((n_seq * n_in, feat_dim)).astype(np.float32)
    in_idx = np.array([(n, t, 0) for n in range(n_seq) for t in range(-left, 3 * T + left)], np.int32)
    ivec = rng.standard_normal((n_seq, ivec_dim)).astype(np.float32)
    iv_idx = np.array([(n, 0, 0) for n in range(n_seq)], np.int32)
    out_idx = np.array([(n, 3 * t, 0) for t in range(T) for n in range(n_seq)], np.int32)  # frame-major
    dw = rng.choice([0.0, 1.0], size=n_seq * T, p=[0.1, 0.9]).astype(np.float32)
    return dict(inputs=[dict(name="input", indexes=in_idx, features=feats), dict(name="ivector", indexes=iv_idx, features=ivec)],
                outputs=[dict(name="output", indexes=out_idx, supervision=sup, deriv_weights=dw)])


def same_fst(a, b):
    return (a.num_states == b.num_states and np.array_equal(a.arc_begin, b.arc_begin) and np.array_equal(a.ilabel, b.ilabel)
            and np.array_equal(a.nextstate, b.nextstate) and np.allclose(a.arc_weight, b.arc_weight)
            and np.array_equal(np.isinf(a.final), np.isinf(b.final)) and np.allclose(a.final[~np.isinf(a.final)], b.final[~np.isinf(b.final)]))


def write_set(tmp_path, fst, lengths, **kwargs):
    keyed = [("utt%03d-%d" % (i, L), make_example(fst, L, seed=20 + i)) for i, L in enumerate(lengths)]
    ark, scp = str(tmp_path / "egs.ark"), str(tmp_path / "egs.scp")
    kw.write_ark(ark, keyed, scp_path=scp, **kwargs)
    return keyed, ark, scp
