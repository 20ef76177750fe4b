"""Generates tests/golden/*.npz: seeded inputs plus expected outputs of the chain objective.

The reference (nttcslab-sp/torchain) cannot be imported or run here (it needs Kaldi, OpenFst, CUDA
and PyTorch 0.4) and holds no golden vectors of its own, so the expected
outputs come from oracle/independent_f64.py -- a float64, log-semiring, autograd formulation that
shares no code with either the C oracle or the HIP kernels.  Inputs are stored as data (FST arrays,
nnet output), so the fixtures stay valid if the generators in torchain_amd/synth.py change.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import independent_f64 as ind  # noqa: E402
from torchain_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, fst, S, T, l2, leaky, sup weight, zero output
    ("c1_small", synth.left_to_right_den_fst(200, seed=42), 4, 12, 5e-5, 1e-5, 1.0, False),
    ("c1_leaky02", synth.left_to_right_den_fst(200, seed=42), 3, 10, 0.0, 0.2, 0.5, False),
    ("rand_graph", synth.random_den_fst(48, 4, 40, seed=2), 4, 11, 1e-3, 0.1, 1.0, False),
    ("skewed_graph", synth.skewed_den_fst(40, 400, 30, seed=3), 3, 9, 0.0, 0.05, 1.0, False),
    ("zero_output", synth.random_den_fst(24, 3, 16, seed=5), 2, 8, 0.0, 1e-5, 1.0, True),
]


def main():
    for name, fst, S, T, l2, leaky, w, zero in CASES:
        pi = synth.initial_probs_f64(fst)
        sup = synth.random_supervision(fst, S, T, 3, seed=7, weight=w, initial_probs=pi)
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=1234, zero=zero)
        r = ind.chain_objf_and_deriv(fst, pi, sup, y, l2, leaky)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            # inputs
            den_num_states=fst.num_states, den_src=fst.src, den_dst=fst.dst, den_ilabel=fst.ilabel,
            den_weight=fst.weight, den_final=fst.final, den_start=fst.start, num_pdfs=fst.num_pdfs,
            sup_weight=sup.weight, num_sequences=S, frames_per_sequence=T, sup_num_states=sup.num_states,
            sup_arc_begin=sup.arc_begin, sup_ilabel=sup.ilabel, sup_arc_weight=sup.arc_weight,
            sup_nextstate=sup.nextstate, sup_final=sup.final, nnet_output=y, l2_regularize=l2, leaky=leaky,
            # expected (float64)
            initial_probs=pi, objf=r["objf"], l2_term=r["l2_term"], weight=r["weight"], num_logprob=r["num"],
            den_logprob=r["den"], deriv=r["deriv"].astype(np.float32), xent_deriv=r["xent_deriv"].astype(np.float32),
            den_deriv=r["den_deriv"].astype(np.float32))
        print("%-14s objf %.6f l2 %.6f weight %g  (%d KB)" % (
            name, r["objf"], r["l2_term"], r["weight"], os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024))


if __name__ == "__main__":
    main()
