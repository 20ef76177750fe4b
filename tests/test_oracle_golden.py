"""The CPU oracle (oracle/chain_oracle.c, Kaldi-style float/double arithmetic) against the committed
golden vectors (tests/golden/*.npz, produced by the independent float64 autograd formulation).
This is what pins the restatement: parity with the reference itself is unpinned (no Kaldi here)."""
import os

import numpy as np
import pytest

from torchain_amd import synth

from fixtures import GOLDEN, load_golden as load


def test_fixtures_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    z, fst, sup = load(path)
    g = oracle.DenGraph(fst)
    np.testing.assert_allclose(g.initial_probs(), z["initial_probs"], rtol=1e-5, atol=1e-9)
    out = oracle.compute_chain_objf_and_deriv(g, sup, z["nnet_output"], float(z["l2_regularize"]), float(z["leaky"]),
                                              want_xent=True)
    assert abs(out["objf"] - float(z["objf"])) <= 1e-5 * abs(float(z["objf"]))
    assert abs(out["l2_term"] - float(z["l2_term"])) <= 1e-5 * abs(float(z["l2_term"])) + 1e-12
    assert out["weight"] == float(z["weight"]) == sup.weight * sup.num_sequences * sup.frames_per_sequence
    scale = max(np.abs(z["deriv"]).max(), sup.weight)
    assert np.abs(out["deriv"] - z["deriv"]).max() <= 1e-5 * scale
    assert np.abs(out["xent_deriv"] - z["xent_deriv"]).max() <= 1e-5 * scale
    den = oracle.den_forward_backward(g, z["nnet_output"], sup.num_sequences, float(z["leaky"]))
    frames = sup.num_sequences * sup.frames_per_sequence  # log-probs scale with the number of frames
    assert abs(den["logprob"] - float(z["den_logprob"])) <= 1e-5 * max(abs(float(z["den_logprob"])), frames)
    assert np.abs(den["deriv"] - z["den_deriv"]).max() <= 1e-5
    num = oracle.num_forward_backward(sup, z["nnet_output"])
    assert abs(num["logprob_weighted"] - sup.weight * float(z["num_logprob"])) <= 1e-5 * max(
        abs(float(z["num_logprob"])), frames)


def test_oracle_matches_independent_formulation_live(oracle):
    """A fresh seed, not in the fixtures: the two formulations are compared directly."""
    from oracle import independent_f64 as ind

    fst = synth.random_den_fst(30, 3, 20, seed=77)
    S, T = 3, 9
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=78, weight=0.5, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, 20, seed=79, scale=2.0)
    a = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-3, 0.1, want_xent=True)
    b = ind.chain_objf_and_deriv(fst, synth.initial_probs_f64(fst), sup, y, 1e-3, 0.1)
    assert abs(a["objf"] - b["objf"]) <= 1e-5 * abs(b["objf"])
    assert np.abs(a["deriv"] - b["deriv"]).max() <= 1e-5
    assert np.abs(a["xent_deriv"] - b["xent_deriv"]).max() <= 1e-5
