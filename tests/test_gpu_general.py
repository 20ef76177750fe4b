"""General (not chain-structured) graphs on chip: the owner-computes kernel (den_general_owner.hip) and round 1's kernel
(den_kernels.hip) against each other, the tied kernels and the oracle; skewed degrees; peaky outputs.  REL = 1e-4."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from torchain_amd import io, synth
from torchain_amd._lib import check, lib

from helpers import (REL, check_full, compare_at_size, elementwise, float64_truth, free_port, from3d, hip_chain, hip_den, hip_num,
                     occupy_half_the_cus, oracle_den, peaky_check, peaky_elem, rel_err, to3d)

pytestmark = pytest.mark.gpu


def test_skewed_graph_row_splitting(oracle):
    """Hub states with hundreds of in/out arcs, arbitrary arc->pdf labels, non-final states:
    exercises virtual-row splitting in the schedule."""
    fst = synth.skewed_den_fst(300, 6000, 120, seed=4)
    check_full(oracle, fst, 4, 15, l2=0.0, leaky=0.1)


def test_tied_and_general_kernels_agree(oracle, kernel_family):
    """The same chain-structured graph through both device code paths: the factorised "tied" kernel
    (exp(y) taken out of the arc sums, gamma from per-state quantities) and the general kernel
    (forced with tc_debug_set("force_general")).  Both must match the oracle; the graph also has states with an
    extra self-loop carrying the forward pdf, parallel arcs and a state without a self-loop."""
    from torchain_amd import io

    base = synth.random_den_fst(300, 6, 150, seed=41)
    src, dst, il, w = (np.array(x) for x in (base.src, base.dst, base.ilabel, base.weight))
    # make state 5's self-loop a forward-class arc (same pdf as its other in-arcs) and drop state 7's self-loop
    keep = np.ones(len(src), bool)
    into5 = (dst == 5) & (src != 5)
    if into5.any():
        il[(src == 5) & (dst == 5)] = il[into5][0]
    keep[(src == 7) & (dst == 7)] = False
    fst = base._replace(src=src[keep], dst=dst[keep], ilabel=il[keep], weight=w[keep])
    S, T = 4, 21
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=42, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=43)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1)
    tied_graph = io.DenominatorGraph(fst, fst.num_pdfs)
    assert tied_graph.stats()["tied"] == 1
    kernel_family("force_general")
    general_graph = io.DenominatorGraph(fst, fst.num_pdfs)
    kernel_family("force_general", 0)
    assert general_graph.stats()["tied"] == 0
    outs = []
    for graph in (tied_graph, general_graph):
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, graph=graph)
        assert abs(out["results"][0] - ref["objf"]) <= REL * abs(ref["objf"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
        outs.append(out["deriv"])
    assert rel_err(outs[0], outs[1], floor=1.0) <= REL


@pytest.mark.parametrize("family", ["force_streamed", "force_general"])
@pytest.mark.parametrize("scale", [5.0, 20.0])
def test_other_kernel_families_peaky(oracle, kernel_family, family, scale):
    kernel_family(family)
    peaky_check(oracle, synth.random_den_fst(300, 5, 100, seed=32), 2, 150, scale, 0.1)


# ---- the general on-chip kernel on owner-computes schedules (den_general_owner.hip) ---------------------------------------
@pytest.mark.parametrize("which", ["forced", "skewed", "three_planes_of_pdfs"])
def test_general_owner_kernel_against_round1_kernel_and_oracle(oracle, kernel_family, which):
    """General graphs of at most 8192 states take round 5's kernel (owner-computes schedules, 8-byte cells, two barriers per
    frame); `old_general` keeps round 1's.  Both against the oracle through the full objective; Kaldi's accumulate form; the
    forward-only call."""
    if which == "forced":
        kernel_family("force_general")
        fst = synth.random_den_fst(5000, 6, 900, seed=21)
    elif which == "skewed":
        fst = synth.skewed_den_fst(1500, 12000, 400, seed=22)
    else:
        kernel_family("force_general")
        fst = synth.random_den_fst(3000, 4, 9000, seed=23)
    S, T = 5, 17
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=31, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=32)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    outs = []
    for old in (0, 1):
        kernel_family("old_general", old)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True)
        assert out["graph"].stats()["tied"] == 0
        res = out["results"]
        assert abs(res[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T), (which, old, res, ref["results"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, (which, old)
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL, (which, old)
        outs.append(out)
    assert not np.array_equal(outs[0]["deriv"], outs[1]["deriv"])  # (they ARE two kernels)
    kernel_family("old_general", 0)
    dref = oracle.den_forward_backward(g, y, S, leaky=0.05, deriv_weight=1.0)
    acc = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert abs(acc["logprob"] - dref["logprob"]) <= REL * abs(dref["logprob"]) and acc["status"] == 0
    assert rel_err(acc["deriv"] - 0.5, dref["deriv"]) <= REL
    again = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert np.array_equal(acc["deriv"], again["deriv"]) and acc["logprob"] == again["logprob"]  # bitwise reproducible
    fwd = hip_den(fst, y, S, leaky=0.05, want_deriv=False)
    assert abs(fwd["logprob"] - dref["logprob"]) <= REL * abs(dref["logprob"])


def test_general_owner_kernel_full_size_and_peaky(oracle, kernel_family):
    """The C3 graph forced onto the general kernel at 64 x 150 against the oracle (element-wise bounds), and a peaky T = 150
    sequence against the float64 formulation (tests/test_gpu_peaky.py's rule)."""
    kernel_family("force_general")
    c = synth.CONFIGS["C3"]
    fst = synth.config_den_fst("C3")
    S, T = 64, 150
    y = synth.random_nnet_output(S, T, c["P"], seed=1241)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 0 and out["status"] == 0
    ref_lp, ref = oracle_den(oracle, fst, y, S, T, c["leaky"])
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "C3 forced general")
    peaky_check(oracle, synth.config_den_fst("C2"), 1, 150, 10.0, 0.1)
