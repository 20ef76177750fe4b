"""A miniature of the recipe's data and model for the end-to-end tests (tests/test_gpu_recipe.py): chain egs whose FEATURES carry
their supervision (so that a small network can learn them in a few dozen steps), written as a Kaldi archive + scp + length file
with tests/kaldi_egs_writer.py, and a two-layer ``Conv1d`` network with the recipe's two heads.  Builder-authored; the call
pattern it serves is ``/root/reference/example/chime5/train_faster.py:117-145`` (nothing of that file is used)."""
import numpy as np
import torch

from torchain_amd import synth

import kaldi_egs_writer as kw

LEFT = 4        # input context either side of the output frames (frame-subsampling 3: 3 T + 2 LEFT input frames)
FEAT, IVEC = 16, 3


def frame_pdfs(sup):
    """pdf ids (0-based) on the arcs leaving the states of every frame of a one-sequence supervision acceptor."""
    times = np.full(sup.num_states, -1, np.int64)
    times[0] = 0
    out = [set() for _ in range(sup.frames_per_sequence)]
    for s in range(sup.num_states):
        for a in range(sup.arc_begin[s], sup.arc_begin[s + 1]):
            times[sup.nextstate[a]] = times[s] + 1
            out[times[s]].add(int(sup.ilabel[a]) - 1)
    return out


def learnable_example(fst, T, seed, emb, noise=0.3):
    """One chain eg of T output frames: a supervision of two random paths through ``fst`` and input features that are the mean
    embedding of the frame's numerator pdfs (+ noise) around input frame 3 t."""
    rng = np.random.default_rng(seed)
    sup = synth.random_supervision(fst, 1, T, 2, seed=seed)
    n_in = 3 * T + 2 * LEFT
    feats = (noise * rng.standard_normal((n_in, FEAT))).astype(np.float32)
    for t, pdfs in enumerate(frame_pdfs(sup)):
        feats[LEFT + 3 * t - 1:LEFT + 3 * t + 2] += emb[sorted(pdfs)].mean(axis=0)
    in_idx = np.array([(0, t, 0) for t in range(-LEFT, 3 * T + LEFT)], np.int32)
    ivec = rng.standard_normal((1, IVEC)).astype(np.float32)
    out_idx = np.array([(0, 3 * t, 0) for t in range(T)], np.int32)
    return dict(inputs=[dict(name="input", indexes=in_idx, features=feats),
                        dict(name="ivector", indexes=np.array([(0, 0, 0)], np.int32), features=ivec)],
                outputs=[dict(name="output", indexes=out_idx, supervision=sup, deriv_weights=np.ones(T, np.float32))])


def write_learnable_set(directory, fst, lengths, seed=0, name="train"):
    """-> scp path (its ``.len`` file beside it): one eg per entry of ``lengths``."""
    emb = np.random.default_rng(1234).standard_normal((fst.num_pdfs, FEAT)).astype(np.float32)
    keyed = [("%s%03d-%d" % (name, i, L), learnable_example(fst, L, seed=seed + i, emb=emb)) for i, L in enumerate(lengths)]
    ark, scp = str(directory / (name + ".ark")), str(directory / (name + ".scp"))
    kw.write_ark(ark, keyed, scp_path=scp)
    with open(scp + ".len", "w") as f:
        f.write("".join("%s %d\n" % (k, L) for (k, _), L in zip(keyed, lengths)))
    return scp


class TwoLayerTdnn(torch.nn.Module):
    """(B, FEAT, 3 T + 2 LEFT) features + (B, IVEC) i-vector -> LF-MMI output and cross-entropy output, both (B, n_pdf, T)."""

    def __init__(self, n_pdf, hidden=64):
        super().__init__()
        self.conv = torch.nn.Conv1d(FEAT, hidden, kernel_size=2 * LEFT + 1, stride=3)
        self.aux = torch.nn.Linear(IVEC, hidden)
        self.mmi = torch.nn.Conv1d(hidden, n_pdf, 1)
        self.xent = torch.nn.Conv1d(hidden, n_pdf, 1)

    def forward(self, feats, ivec):
        h = torch.relu(self.conv(feats) + self.aux(ivec).unsqueeze(2))
        return self.mmi(h), self.xent(h)
