"""Utterances too long for the LDS a large graph leaves: the frame sums asum_0..T, round4(T + 1) floats behind
the on-chip layout, no longer fit -- the plane-wise kernel of 7-plane graphs above T ~ 1000, the general owner-computes kernel of
8192 states x ~7800 pdfs above T ~ 300.  Until round 6 such a launch returned TC_ERR_UNSUPPORTED although the graph had built,
where the kernels these two replaced (streamed path, round 1's general kernel) ran any T.  Now the sums go through the workspace
(DenLayout::asum_global); compared with the oracle here, with the library's counter showing that this form ran.
Reference: any T is legal for ``src/my_lib_chain.cpp:129-131`` ([K] DenominatorComputation has no length limit)."""
import numpy as np
import pytest

from torchain_amd import synth
from torchain_amd._lib import lib

from helpers import hip_den, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


def _long_launches():
    return int(lib.tc_debug_counter(b"den_long_utterance_launches"))


def _check(oracle, fst, S, T, expect_tied, expect_long, want_deriv=True):
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=T)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    before = _long_launches()
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, want_deriv=want_deriv)
    assert (_long_launches() - before > 0) == expect_long
    assert out["graph"].stats()["tied"] == expect_tied
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    if want_deriv:
        assert out["status"] == 0
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
        assert np.abs(out["deriv"].sum(axis=1, dtype=np.float64) - 1.0).max() <= 1e-4
    return out


def test_seven_plane_graph_long_utterance(oracle):
    fst = synth.random_den_fst(28000, 3, 2928, seed=61)
    _check(oracle, fst, 2, 1300, expect_tied=1, expect_long=True)
    _check(oracle, fst, 2, 1300, expect_tied=1, expect_long=True, want_deriv=False)
    _check(oracle, fst, 2, 40, expect_tied=1, expect_long=False)  # the same graph, sums in LDS


def test_general_graph_near_the_lds_limit_long_utterance(oracle, kernel_family):
    kernel_family("force_general")
    fst = synth.random_den_fst(8000, 3, 7800, seed=62)
    _check(oracle, fst, 2, 700, expect_tied=0, expect_long=True)
    _check(oracle, fst, 2, 100, expect_tied=0, expect_long=False)
