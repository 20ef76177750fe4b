"""The on-chip kernels of TIED (chain-structured) graphs of at most 16384 positions -- the metric's kernel family: the fused
kernel (den_tied_kernel.hip on den_tied_frames.h), its two-CU forms for small batches (den_tied_split.hip, den_tied_mitm.hip),
the two-sequence kernel (den_tied_pair.hip) and the per-graph choice between them -- against the oracle at full C2 / C3 / C5 /
R1-R3 sizes, element-wise against float64, on peaky outputs, with hub states / partial planes / split states, under HIP-graph
capture, with a co-tenant, bitwise reproducible.  [K] DenominatorComputation via ``src/my_lib_chain.cpp:129-131``.  REL = 1e-4."""
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


def test_tied_graph_with_hub_states(oracle):
    """Chain-structured graph whose hub states have arc lists far longer than one schedule row: the
    owner-computes schedules split them into secondary rows that other lanes walk and the owner folds
    in after a barrier; states without a self-loop and non-final states as well."""
    fst = synth.skewed_tied_den_fst(400, 7000, 150, seed=8)
    from torchain_amd import io
    g = io.DenominatorGraph(fst, fst.num_pdfs)
    assert g.stats()["tied"] == 1
    indeg = np.bincount(fst.dst[fst.src != fst.dst], minlength=400)
    assert indeg.max() > 64  # really needs secondary rows
    check_full(oracle, fst, 4, 15, l2=1e-4, leaky=0.05)


def test_tied_graph_partial_planes(oracle):
    """State counts that do not fill the 4096-position planes of the tied layout (phantom positions):
    one partly filled plane, and one full plane plus a partly filled one."""
    check_full(oracle, synth.random_den_fst(1500, 3, 257, seed=21), 2, 7, l2=0.0, leaky=0.1)
    check_full(oracle, synth.random_den_fst(5000, 3, 300, seed=22), 2, 5, l2=0.0, leaky=0.1)
    # three and four planes: the <JV=4> instantiations (8193..16384 states), small and mid vocabularies
    from torchain_amd import io
    for H, P, seed in ((9000, 5000, 23), (14000, 2000, 24)):
        fst = synth.random_den_fst(H, 3, P, seed=seed)
        assert io.DenominatorGraph(fst, P).stats()["tied"] == 1
        check_full(oracle, fst, 2, 4, l2=1e-4, leaky=0.1)


def test_tied_tight_layout_mid_vocab(oracle):
    """4097..8192 pdfs with 8192 states: the tied kernel's roomy LDS layout does not fit, the tight one
    (alpha' re-read from the history, exp(y) rewritten in place) does; 6000 pdfs with fewer states fits
    the roomy layout of the same <JV=2, PV=2> instantiation."""
    from torchain_amd import io
    fst = synth.random_den_fst(8192, 3, 6000, seed=41)
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1
    check_full(oracle, fst, 2, 9, l2=1e-4, leaky=0.1)
    check_full(oracle, synth.random_den_fst(3000, 4, 6000, seed=42), 3, 7, l2=0.0, leaky=0.05)


def test_nearly_tied_graph_state_splitting(oracle):
    """States entered through several pdfs are split into one copy per pdf (exact) so that the graph stays
    on the tied kernel; compared with the oracle run on the ORIGINAL graph."""
    from torchain_amd import io
    fst = synth.nearly_tied_den_fst(700, 5, 150, seed=15, fraction=0.05)
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1
    check_full(oracle, fst, 3, 12, l2=1e-4, leaky=0.1)
    check_full(oracle, synth.nearly_tied_den_fst(64, 4, 20, seed=6, fraction=0.3), 2, 9, l2=0.0, leaky=1e-5)


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_denominator_alone_and_accumulate(oracle, kernel_family, form):
    """[K] DenominatorComputation used directly (chain-supervision-test.hpp:403-423): log-prob,
    Backward(1.0, &deriv) semantics (adds into deriv), sum(deriv) = S*T."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(200, 6, 90, seed=11)
    S, T = 5, 19
    g = oracle.DenGraph(fst)
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=2)
    ref = oracle.den_forward_backward(g, y, S, leaky=1e-5, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=1e-5, deriv_weight=1.0, accumulate=True, init=0.25)
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert out["status"] == 0 and ref["ok"]
    assert rel_err(out["deriv"] - 0.25, ref["deriv"]) <= REL
    assert abs(out["deriv"].sum() - 0.25 * y.size - S * T) < 10.0 * 1e-2
    # overwrite form with the fused l2 term
    out2 = hip_den(fst, y, S, leaky=1e-5, deriv_weight=-0.5, l2_scale=1e-3, accumulate=False)
    assert rel_err(out2["deriv"], -0.5 * ref["deriv"] - 1e-3 * y) <= REL
    # forward only
    out3 = hip_den(fst, y, S, leaky=1e-5, want_deriv=False)
    assert abs(out3["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])


def test_two_cu_form_agrees_with_fused_kernel(oracle, kernel_family):
    """Small batches of tied graphs run forward and backward recursion side by side on two CUs and form gamma in a
    third pass (den_tied_split.hip).  Same results as the fused kernel -- to rounding: the backward recursion keeps
    normalisers of its own -- on plain tied graphs, graphs with hub states (secondary rows) and nearly tied graphs
    (split states), overwrite and accumulate forms, one frame and many."""
    cases = [(synth.config_den_fst("C2"), 5, 40, 0.1), (synth.config_den_fst("C2"), 2, 1, 1e-5),
             (synth.skewed_tied_den_fst(600, 6000, 300, seed=5), 3, 25, 0.05),
             (synth.nearly_tied_den_fst(900, 5, 400, seed=8, fraction=0.2), 4, 33, 0.2),
             (synth.random_den_fst(5000, 4, 6000, seed=3), 2, 12, 0.1),
             (synth.config_den_fst("C5"), 2, 12, 0.1),                  # the fused kernel's tight layout
             (synth.random_den_fst(9000, 3, 5000, seed=23), 2, 9, 1e-5)]  # 16 states per thread
    for fst, S, T, leaky in cases:
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=21, scale=2.0)
        ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
        outs = {}
        for form in ("two_cu", "fused"):
            kernel_family("no_phase_split", 1 if form == "fused" else 0)
            a = hip_den(fst, y, S, leaky=leaky, deriv_weight=-1.0, l2_scale=1e-3)
            b = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, accumulate=True, init=0.5, graph=a["graph"])
            assert a["status"] == 0 and b["status"] == 0
            assert abs(a["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
            assert rel_err(a["deriv"], -ref["deriv"] - 1e-3 * y) <= REL
            assert rel_err(b["deriv"] - 0.5, ref["deriv"]) <= REL
            outs[form] = a
        assert outs["two_cu"]["logprob"] == outs["fused"]["logprob"]  # the forward recursion is the same code
        assert np.abs(outs["two_cu"]["deriv"] - outs["fused"]["deriv"]).max() <= 2e-6


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_phone_lm_structured_graphs(oracle, kernel_family, form):
    """Graphs with the structure of Kaldi's chain den.fst (synth.phone_lm_den_fst: pruned phone LM x one-state
    chain topology x biphone tree; in-degrees from 1 to hundreds, popular back-off states): a small one, and the
    13800-state one whose secondary rows only fit the LDS once the home rows are allowed to grow (den_graph.cpp:
    build_schedules) -- it must stay on the on-chip kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    for fst, S, T in ((synth.phone_lm_den_fst(num_histories=90, branching=9, num_pdfs=400, seed=3), 4, 30),
                      # with the LM's empty history: its phone instances are entered through arcs of up to 42 pdfs and
                      # are split into as many copies (schedule_owner.cpp: make_work_graph)
                      (synth.phone_lm_den_fst(num_histories=600, branching=8, num_pdfs=900, seed=3, unigram_fraction=0.05), 2, 10),
                      (synth.config_den_fst("R2"), 2, 8)):
        graph = io.DenominatorGraph(fst, fst.num_pdfs)
        assert graph.stats()["tied"] == 1
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=4, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=5, scale=2.0)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True, graph=graph)
        assert abs(out["results"][0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T)
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_long_sequences(oracle, kernel_family, form):
    """1000 frames per sequence (the per-frame normalisers live in LDS; the two-CU form chains 1000 of them in
    double): objective and derivatives against the oracle."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(300, 5, 120, seed=29)
    S, T = 2, 1000
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=2, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=3, scale=2.0)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.05, want_xent=False)
    out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.05)
    assert abs(out["results"][0] - ref["objf"]) <= REL * abs(ref["objf"])
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_heavily_split_graphs(oracle, kernel_family, form):
    """Graphs in which most states are entered through two or three pdfs are still run on the tied kernel, with up to
    2.5x the states after splitting (schedule_owner.cpp: make_work_graph); beyond that they take the general kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    for frac, want_tied in ((0.6, 1), (0.9, 1)):
        fst = synth.nearly_tied_den_fst(1200, 6, 500, seed=31, fraction=frac)
        graph = io.DenominatorGraph(fst, fst.num_pdfs)
        assert graph.stats()["tied"] == want_tied and graph.stats()["bwd_rows"] > 1.5 * fst.num_states
        S, T = 3, 25
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=4, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=5, scale=2.0)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=False)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, graph=graph)
        assert abs(out["results"][0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T)
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL


def test_full_size_properties_config3():
    """BASELINE.json configs[2] at full size (S=256, T=150, P=4096): size-independent properties
    the reference's own test asserts (chain-supervision-test.hpp:417-423,267-283): gamma sums to
    one per (frame, sequence); row sums of the full derivative vanish; objf <= 0."""
    c = synth.CONFIGS["C3"]
    fst = synth.config_den_fst("C3")
    S, T, P = c["S"], c["T"], c["P"]
    y = synth.random_nnet_output(S, T, P, seed=1237)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0, accumulate=False)
    assert out["status"] == 0
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() < 1e-3
    assert abs(out["deriv"].sum(dtype=np.float64) - S * T) < 10.0
    assert out["deriv"].min() >= 0.0
    # shift property of the denominator: adding r[row] to every pdf of a row adds sum(r) to the log-prob
    r = np.random.default_rng(0).standard_normal(S * T).astype(np.float32)
    out_s = hip_den(fst, y + r[:, None], S, leaky=c["leaky"], want_deriv=False, graph=out["graph"])
    assert abs((out_s["logprob"] - out["logprob"]) - float(r.sum(dtype=np.float64))) < 1e-4 * abs(out["logprob"])
    # full objective with a numerator that is a weighted subset of denominator paths
    sup = synth.random_supervision(fst, S, T, 3, seed=7, initial_probs=out["graph"].initial_probs())
    full = hip_chain(fst, sup, y, l2=0.0, leaky=c["leaky"], graph=out["graph"])
    assert full["results"][0] <= 0.0
    assert full["results"][2] == S * T
    rs = full["deriv"].sum(axis=1, dtype=np.float64)
    assert np.linalg.norm(rs) < 0.1 and abs(full["deriv"].sum(dtype=np.float64)) < 0.2


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_hot_call_can_be_captured_in_a_hip_graph(kernel_family, form):
    """include/torchain_hip.h promises no allocation and no host synchronisation in the hot calls: then a training
    loop may capture them in a HIP graph (stream capture forbids both) and replay it on new data.  The two-CU form
    forks to a side stream and joins again inside the call, which capture follows."""
    import ctypes as C
    from torchain_amd._lib import check, lib
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("C2")
    S, T, P = 4, 20, fst.num_pdfs
    graph = io.DenominatorGraph(fst, P).prepare(torch.device("cuda", 0))
    y = torch.randn(S * T, P, device="cuda")
    deriv = torch.zeros_like(y)
    lp = torch.zeros(1, dtype=torch.float64, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def call():
        check(lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), 0.1, -1.0, 1e-4, 0,
            C.c_void_p(deriv.data_ptr()), deriv.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(st.data_ptr()),
            C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "den")

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()  # warm-up outside the capture: per-device tables, kernel attributes, side stream
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call()
    for seed in (1, 2):
        y.copy_(torch.randn(S * T, P, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)))
        g.replay()
        torch.cuda.synchronize()
        got, got_lp = deriv.clone(), float(lp)
        call()
        torch.cuda.synchronize()
        assert int(st) == 0 and float(lp) == got_lp
        assert torch.equal(deriv, got)


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_results_are_bitwise_reproducible(kernel_family, form):
    """gamma is accumulated in integer fixed point and every float sum has a fixed order, so two runs on
    the same inputs give identical bits (Kaldi's float atomics do not)."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("C2")
    S, T = 8, 30
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=3)
    a = hip_den(fst, y, S, leaky=0.1, deriv_weight=-1.0, l2_scale=1e-4)
    b = hip_den(fst, y, S, leaky=0.1, deriv_weight=-1.0, l2_scale=1e-4, graph=a["graph"])
    assert a["logprob"] == b["logprob"]
    assert np.array_equal(a["deriv"], b["deriv"])


@pytest.mark.parametrize("leaky", [0.1, 1e-5])
def test_config3_full_size_denominator(oracle, leaky):
    """configs[2], the metric's workload: batch 256 x 150 frames x 4096 pdfs; log-prob and the whole derivative.
    leaky = 1e-5 is the API default (torchain/functions.py:128-130) at the full 150 frames."""
    c = synth.CONFIGS["C3"]
    fst = synth.config_den_fst("C3")
    S, T, P = c["S"], c["T"], c["P"]
    y = synth.random_nnet_output(S, T, P, seed=1237)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
    assert out["status"] == 0 and ref["ok"]
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"]) <= REL
    # element-wise on everything that is not rounding dust: the per-state (subtraction) form of gamma on tied graphs
    big = ref["deriv"] > 1e-4
    assert (np.abs(out["deriv"][big] - ref["deriv"][big]) / ref["deriv"][big]).max() <= 1e-3


def test_config5_full_size_denominator(oracle):
    """configs[4]: large-vocabulary graph (P=10240, A=61440), batch 128 x 150, leaky 0.1: the tight LDS layout."""
    c = synth.CONFIGS["C5"]
    fst = synth.config_den_fst("C5")
    S, T, P = c["S"], c["T"], c["P"]
    y = synth.random_nnet_output(S, T, P, seed=1239)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=c["leaky"], deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 1
    assert out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"]) <= REL


@pytest.mark.parametrize("form", ["two_cu", "fused", "meet_in_the_middle", "two_sequence"])
@pytest.mark.parametrize("leaky", [1e-5, 0.1])
@pytest.mark.parametrize("scale", [5.0, 10.0, 20.0])
def test_tied_kernel_peaky_outputs_t150(oracle, kernel_family, scale, leaky, form):
    """CHiME5-like graph (the C2 / C3 graph, tied kernel, roomy layout), T = 150, API-default and config leaky;
    as a small batch runs it (forward and backward recursion on two CUs, den_tied_split.hip) and as the fused
    kernel that batches beyond half the chip take."""
    if form == "fused":
        kernel_family("no_phase_split")
    elif form == "meet_in_the_middle":  # (round 3: den_tied_mitm.hip, the default from 32 sequences on)
        kernel_family("force_mitm")
    elif form == "two_sequence":        # (round 3: den_tied_pair.hip; one sequence = a pair with a phantom partner)
        kernel_family("force_pair")
    peaky_check(oracle, synth.config_den_fst("C2"), 1, 150, scale, leaky)


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_rows_beyond_the_exp_clamp(oracle, kernel_family, form):
    if form == "fused":
        kernel_family("no_phase_split")
    peaky_check(oracle, synth.config_den_fst("C2"), 1, 60, 10.0, 1e-5, beyond_clamp=True)


@pytest.mark.parametrize("scale", [5.0, 20.0])
def test_tight_layout_and_jv4_peaky(oracle, scale):
    """The tight LDS layout (C5 graph: alpha re-read from the history) and the 16-states-per-thread instantiation."""
    peaky_check(oracle, synth.config_den_fst("C5"), 1, 60, scale, 1e-5)
    peaky_check(oracle, synth.random_den_fst(9000, 3, 5000, seed=23), 1, 40, scale, 0.1)


@pytest.mark.parametrize("name,S", [("R1", 64), ("R3", 16), ("R1", 161)])
def test_phone_lm_graphs_at_full_size(oracle, name, S):
    """Graphs with the structure of a real chain den.fst (pruned phone LM x topology x tree; in-degrees 1 .. ~130): R1
    at configs[1]'s batch of 64 x 150 frames, R3 -- whose empty-history states the library splits into 9681
    chain-structured ones (12 states per thread) -- at 16 x 150, and R1 at an odd batch above one sequence per two CUs
    (161 x 150: the kernel the library's own timing chose for the graph -- on an MI355X the two-sequence one --
    with the numerator beside it); full objective vs the oracle."""
    c = synth.CONFIGS[name]
    fst = synth.config_den_fst(name)
    T, P = c["T"], c["P"]
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=9, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, P, seed=1240)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, c["l2"], c["leaky"], want_xent=True)
    out = hip_chain(fst, sup, y, l2=c["l2"], leaky=c["leaky"], xent=True)
    assert out["graph"].stats()["tied"] == 1
    res = out["results"]
    assert abs(res[0] - ref["objf"]) <= REL * abs(ref["objf"]), (res, ref["results"])
    assert abs(res[1] - ref["l2_term"]) <= REL * abs(ref["l2_term"])
    assert res[2] == ref["weight"] == S * T
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL


# ---- the two-sequence kernel -------------------------------------------------------------------------------
def _pair_vs_fused(oracle, kernel_family, fst, S, T, leaky, seed, l2=0.0, accumulate=False, with_oracle=True):
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed)
    kernel_family("no_pair")
    kernel_family("no_phase_split")
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    a = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    kernel_family("no_pair", 0)
    kernel_family("force_pair")
    b = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    c = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    assert a["status"] == 0 and b["status"] == 0
    assert b["logprob"] == c["logprob"] and np.array_equal(b["deriv"], c["deriv"])  # reproducible bit for bit
    assert abs(a["logprob"] - b["logprob"]) <= 1e-6 * abs(a["logprob"])
    assert rel_err(b["deriv"], a["deriv"]) <= 2e-5
    if with_oracle:
        ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
        want = ref["deriv"] - l2 * y + (0.25 if accumulate else 0.0)
        assert abs(b["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
        assert rel_err(b["deriv"], want) <= REL
        rows = b["deriv"] + l2 * y - (0.25 if accumulate else 0.0)
        assert np.abs(rows.sum(axis=1, dtype=np.float64) - 1.0).max() <= 1e-4  # [K]: sum_pdf gamma_t = 1


@pytest.mark.parametrize("case", ["even", "odd", "two_frames", "one_sequence", "accumulate_l2"])
def test_two_sequence_kernel_small_graph(oracle, kernel_family, case):
    fst = synth.random_den_fst(256, 6, 100, seed=5)
    S, T, leaky, kw = {"even": (4, 20, 0.1, {}), "odd": (5, 7, 1e-5, {}), "two_frames": (2, 2, 0.1, {}),
                       "one_sequence": (1, 3, 0.1, {}), "accumulate_l2": (6, 11, 0.1, dict(l2=5e-5, accumulate=True))}[case]
    _pair_vs_fused(oracle, kernel_family, fst, S, T, leaky, seed=1, **kw)


def test_two_sequence_kernel_hub_states_and_two_planes(oracle, kernel_family):
    """R1 (phone-LM structure: secondary rows of hub states go through LDS slots, folded by the owner lane) and a
    3000-state graph (one plane of positions: the second plane's loads fall outside their descriptors)."""
    _pair_vs_fused(oracle, kernel_family, synth.config_den_fst("R1"), 6, 20, 0.1, seed=7)
    _pair_vs_fused(oracle, kernel_family, synth.random_den_fst(3000, 8, 1500, seed=6), 7, 30, 0.1, seed=6)


@pytest.mark.parametrize("leaky", [0.1, 1e-5])
def test_two_sequence_kernel_c3_graph_t150(oracle, kernel_family, leaky):
    _pair_vs_fused(oracle, kernel_family, synth.config_den_fst("C3"), 9, 150, leaky, seed=8)


def test_two_sequence_kernel_more_workgroups_than_cus(oracle, kernel_family):
    """300 sequences = 300 workgroups on 256 CUs: pairs are formed by ticket, so the partner of a running workgroup is
    always one that has started or is the next to start."""
    _pair_vs_fused(oracle, kernel_family, synth.config_den_fst("C3"), 300, 40, 0.1, seed=10, with_oracle=False)


def test_kernel_choice_is_timed_per_graph(oracle, kernel_family):
    """tc_den_graph_tuning: a graph the two-sequence kernel fits is timed with both kernels when it reaches the device
    and keeps the two-sequence one only when that is at least 3% faster; a batch above one sequence per two CUs then
    agrees with the oracle whichever kernel it ran on.  ``no_tune`` keeps the fused kernel without timing."""
    dense = synth.random_den_fst(8192, 14, 4096, seed=3)  # 14 arcs per state: where the shared walk pays
    graph = io.DenominatorGraph(dense, 4096).prepare(0)
    t = graph.tuning(0)
    assert t["fused_ms"] > 0 and t["two_sequence_ms"] > 0
    assert t["two_sequence_kernel"] == int(t["two_sequence_ms"] < 0.97 * t["fused_ms"])
    S, T = 130, 5
    y = synth.random_nnet_output(S, T, 4096, seed=12)
    got = hip_den(dense, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    ref = oracle.den_forward_backward(oracle.DenGraph(dense), y, S, leaky=0.1, deriv_weight=1.0)
    assert got["status"] == 0
    assert abs(got["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(got["deriv"], ref["deriv"]) <= REL
    kernel_family("no_tune")
    untimed = io.DenominatorGraph(dense, 4096).prepare(0).tuning(0)
    assert untimed == {"two_sequence_kernel": 0, "fused_ms": 0.0, "two_sequence_ms": 0.0}


def test_two_sequence_kernel_repeated_launches_are_identical(kernel_family):
    """200 back-to-back launches of the two-sequence kernel at a batch that fills the chip with pairs (R1 graph, 254
    sequences, short utterances): every launch pairs its workgroups anew by ticket and hands over between CUs once;
    all results are bit-identical and no launch reports a failed hand-over."""
    fst = synth.config_den_fst("R1")
    P = synth.CONFIGS["R1"]["P"]
    S, T = 254, 12
    y = synth.random_nnet_output(S, T, P, seed=77)
    kernel_family("force_pair")
    graph = io.DenominatorGraph(fst, P)
    first = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    assert first["status"] == 0 and np.isfinite(first["logprob"])
    for _ in range(200):
        again = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
        assert again["status"] == 0 and again["logprob"] == first["logprob"]
        assert np.array_equal(again["deriv"], first["deriv"])


# ---- two CUs per sequence meeting in the middle ---------------------------------------------------------------
def _mitm_vs_fused(oracle, kernel_family, fst, S, T, leaky, seed, l2=0.0, accumulate=False, with_oracle=True):
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed)
    kernel_family("no_phase_split")
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    a = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    kernel_family("no_phase_split", 0)
    kernel_family("force_mitm")
    b = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    c = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    kernel_family("force_mitm", 0)
    kernel_family("no_mitm")
    d = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    kernel_family("no_mitm", 0)
    assert a["status"] == 0 and b["status"] == 0 and d["status"] == 0
    assert b["logprob"] == c["logprob"] and np.array_equal(b["deriv"], c["deriv"])  # reproducible bit for bit
    assert abs(a["logprob"] - b["logprob"]) <= 1e-6 * abs(a["logprob"])
    assert rel_err(b["deriv"], a["deriv"]) <= 2e-5 and rel_err(d["deriv"], a["deriv"]) <= 2e-5  # (d: the two-pass form)
    if with_oracle:
        ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
        want = ref["deriv"] - l2 * y + (0.25 if accumulate else 0.0)
        assert abs(b["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
        assert rel_err(b["deriv"], want) <= REL
        rows = b["deriv"] + l2 * y - (0.25 if accumulate else 0.0)
        assert np.abs(rows.sum(axis=1, dtype=np.float64) - 1.0).max() <= 1e-4  # [K]: sum_pdf gamma_t = 1


@pytest.mark.parametrize("case", ["even", "odd_frames", "two_frames", "three_frames", "accumulate_l2"])
def test_meet_in_the_middle_small_graph(oracle, kernel_family, case):
    """den_tied_mitm.hip: the two roles of a sequence on two CUs, one hand-over at T/2 (T = 2: one frame each side)."""
    fst = synth.random_den_fst(256, 6, 100, seed=5)
    S, T, leaky, kw = {"even": (4, 20, 0.1, {}), "odd_frames": (5, 7, 1e-5, {}), "two_frames": (2, 2, 0.1, {}),
                       "three_frames": (1, 3, 0.1, {}), "accumulate_l2": (6, 11, 0.1, dict(l2=5e-5, accumulate=True))}[case]
    _mitm_vs_fused(oracle, kernel_family, fst, S, T, leaky, seed=1, **kw)


@pytest.mark.parametrize("name,S,T", [("R1", 6, 20), ("C3", 8, 150), ("C5", 4, 40), ("R3", 3, 20), ("X1", 2, 12)])
def test_meet_in_the_middle_every_layout(oracle, kernel_family, name, S, T):
    """Hub states (R1), the metric's graph at its full length (C3), three planes of pdfs (C5), 12 and 16 states per
    thread (R3, X1: tight LDS layout)."""
    _mitm_vs_fused(oracle, kernel_family, synth.config_den_fst(name), S, T, synth.CONFIGS[name]["leaky"], seed=8,
                   with_oracle=name != "X1")


def test_meet_in_the_middle_is_the_default_for_larger_batches(oracle, kernel_family):
    """Batches from 24 (graphs of the C3 class) / 48 / 64 sequences up to half the CUs take it by default: 128 x 30 of the C3 graph through the default path equals
    the forced form bit for bit and the two-pass form to 2e-5."""
    fst = synth.config_den_fst("C3")
    S, T = 128, 30
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=14)
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    default = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    kernel_family("force_mitm")
    forced = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    kernel_family("force_mitm", 0)
    kernel_family("no_mitm")
    two_pass = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    assert default["status"] == 0 and np.array_equal(default["deriv"], forced["deriv"])
    assert not np.array_equal(default["deriv"], two_pass["deriv"]) and rel_err(default["deriv"], two_pass["deriv"]) <= 2e-5


@pytest.mark.parametrize("form", ["fused", "meet_in_the_middle", "two_sequence", "two_pass"])
def test_long_utterances(oracle, kernel_family, form):
    """700 frames (the per-frame normalisers live in LDS, the scale chains of the two-CU forms run over hundreds of
    frames): every kernel family of tied on-chip graphs against the oracle."""
    kernel_family({"fused": "no_phase_split", "meet_in_the_middle": "force_mitm", "two_sequence": "force_pair",
                   "two_pass": "no_mitm"}[form])
    fst = synth.config_den_fst("C3")
    S, T = 3, 700
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=3)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"]) <= REL


def test_kernel_choice_is_reproducible_and_can_be_fixed(oracle, tmp_path, monkeypatch):
    """R1 (the graph whose choice the timing makes: the two-sequence kernel wins by 6-12 %)."""
    monkeypatch.setenv("TORCHAIN_TUNING_CACHE", str(tmp_path / "tuning.json"))  # (an empty cache: the first prepare times)
    fst = synth.config_den_fst("R1")
    P = synth.CONFIGS["R1"]["P"]
    S, T = 192, 12
    y = synth.random_nnet_output(S, T, P, seed=31)
    g1 = io.DenominatorGraph(fst, P).prepare("cuda:0")
    t1 = g1.tuning("cuda:0")
    a = hip_den(fst, y, S, leaky=0.1, graph=g1)
    # a fresh handle of the same graph: the cached choice, no timing launches (both times reported as zero)
    g2 = io.DenominatorGraph(fst, P).prepare("cuda:0")
    t2 = g2.tuning("cuda:0")
    assert t2["two_sequence_kernel"] == t1["two_sequence_kernel"]
    assert t1["fused_ms"] > 0 and t2["fused_ms"] == 0.0 and t2["two_sequence_ms"] == 0.0, (t1, t2)
    b = hip_den(fst, y, S, leaky=0.1, graph=g2)
    assert a["logprob"] == b["logprob"] and np.array_equal(a["deriv"], b["deriv"])
    # the other kernel, fixed by the caller on a third handle before it reaches the device: no timing either
    other = 1 - t1["two_sequence_kernel"]
    g3 = io.DenominatorGraph(fst, P).prepare("cuda:0", variant=other)
    t3 = g3.tuning("cuda:0")
    assert t3["two_sequence_kernel"] == other and t3["fused_ms"] == 0.0
    c = hip_den(fst, y, S, leaky=0.1, graph=g3)
    assert not np.array_equal(a["deriv"], c["deriv"])  # (it IS another kernel)
    assert rel_err(c["deriv"], a["deriv"], floor=1.0) <= 2e-6 and abs(c["logprob"] - a["logprob"]) <= 1e-6 * abs(a["logprob"])
    # and switched on a graph that is already on the device: from the next launch on
    check(lib.tc_den_graph_set_variant(g3.ptr, 0, t1["two_sequence_kernel"]), "tc_den_graph_set_variant")
    d = hip_den(fst, y, S, leaky=0.1, graph=g3)
    assert np.array_equal(a["deriv"], d["deriv"])
    g = oracle.DenGraph(fst)
    ref = oracle.den_forward_backward(g, y, S, 0.1, 1.0)
    for out in (a, c):
        assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    assert lib.tc_den_graph_set_variant(g3.ptr, 0, 2) < 0 and lib.tc_den_graph_set_variant(None, 0, 0) < 0


@pytest.mark.parametrize("cfg,S,T", [("C2", 64, 150), ("C5", 128, 150), ("R1", 64, 150), ("R3", 16, 150)])
def test_full_size_derivative_element_wise(oracle, cfg, S, T):
    """The denominator's derivative (posteriors in [0, 1]) entry by entry against the oracle."""
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    y = synth.random_nnet_output(S, T, c["P"], seed=77)
    out = hip_den(fst, y, S, leaky=c["leaky"])
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, c["leaky"], 1.0)
    assert out["status"] == 0 and abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    elementwise(out["deriv"], ref["deriv"], cfg)


def test_golden_derivative_element_wise():
    """... and against the float64 fixtures (tests/golden: generated by oracle/independent_f64.py).  These are SMALL graphs
    (3 to 40 states): a single state carries a probability mass of order one, and the tied kernels form the occupation
    of a state's forward-class arcs by subtraction, alpha_{t+1}(g) - (self-loop part), in float32 -- an absolute error of
    ~1e-7 whatever the difference comes to.  Measured here: 1.25e-4 relative on an entry of 1e-3 (c1_leaky02; the
    Kaldi-style float32 oracle is 7e-7 from the same fixture).  So the per-element bound claimed on such graphs is
    2e-4 above 1e-3; on the 8192-state graphs of the full-size tests above, where no state holds more than ~1e-3 of
    the mass, 1e-4 holds."""
    import os

    from fixtures import GOLDEN, load_golden as load
    checked = 0
    for path in GOLDEN:
        z, fst, sup = load(path)
        y = np.ascontiguousarray(z["nnet_output"], np.float32)
        out = hip_den(fst, y, sup.num_sequences, leaky=float(z["leaky"]))
        ref = np.asarray(z["den_deriv"], np.float64)
        if (np.abs(ref) > 1e-3).any() and (np.abs(ref) > 1e-4).any():
            elementwise(out["deriv"], ref, os.path.basename(path), bounds=((1e-3, 2e-4), (1e-4, 1e-3)))
            assert np.abs(out["deriv"] - ref).max() <= 2e-6  # (absolute: a few float32 ulps of posteriors in [0, 1])
            checked += 1
    assert checked > 0


@pytest.mark.parametrize("mode", ["force_mitm", "force_pair"])
def test_paired_workgroups_with_a_co_tenant(kernel_family, mode):
    """den_tied_mitm.hip / den_tied_pair.hip pair workgroups by ticket and hand rows over through flags in global memory
    (bounded spins, soft failure).  Here another stream keeps the GPU busy with large GEMMs meanwhile: the pairs'
    workgroups are no longer co-resident by default.  No hand-over may fail (status 0, finite log-prob) and the results
    must equal an undisturbed run's bit for bit."""
    kernel_family("no_tune")
    kernel_family(mode)
    fst = synth.config_den_fst("C2")
    P = synth.CONFIGS["C2"]["P"]
    S, T = (96, 60) if mode == "force_mitm" else (200, 60)
    y = synth.random_nnet_output(S, T, P, seed=5)
    graph = io.DenominatorGraph(fst, P)
    quiet = hip_den(fst, y, S, leaky=0.1, graph=graph)
    assert quiet["status"] == 0 and np.isfinite(quiet["logprob"])
    side = torch.cuda.Stream()
    for rep in range(3):
        busy = occupy_half_the_cus(side, 40)
        out = hip_den(fst, y, S, leaky=0.1, graph=graph)
        side.synchronize()
        assert out["status"] == 0 and out["logprob"] == quiet["logprob"], (mode, rep)
        assert np.array_equal(out["deriv"], quiet["deriv"]), (mode, rep)
        del busy


def test_kernel_choice_cache_below_the_python_layer(tmp_path, monkeypatch):
    """a caller of the C ABI alone -- tc_den_graph_create + tc_den_graph_prepare through ctypes, no
    io.DenominatorGraph -- gets the measured kernel choice of an earlier handle from the library's cache without any timing
    launch (both times reported as zero), and the file holds the entry under hash + device name."""
    import ctypes as C
    import json

    import torch
    from torchain_amd._lib import check, lib
    path = tmp_path / "tuning.json"
    monkeypatch.setenv("TORCHAIN_TUNING_CACHE", str(path))
    fst = synth.config_den_fst("R1")
    P = synth.CONFIGS["R1"]["P"]

    def make():
        h = C.c_void_p()
        src, dst, il = (np.ascontiguousarray(a, np.int32) for a in (fst.src, fst.dst, fst.ilabel))
        w, fin = (np.ascontiguousarray(a, np.float32) for a in (fst.weight, fst.final))
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        check(lib.tc_den_graph_create(C.byref(h), int(fst.num_states), len(src), p(src), p(dst), p(il), p(w), p(fin), int(fst.start), P),
              "tc_den_graph_create")
        return h

    def tuning(h):
        choice, a, b = C.c_int32(-1), C.c_float(-1), C.c_float(-1)
        check(lib.tc_den_graph_tuning(h, 0, C.byref(choice), C.byref(a), C.byref(b)), "tc_den_graph_tuning")
        return choice.value, a.value, b.value

    g1 = make()
    check(lib.tc_den_graph_prepare(g1, 0), "tc_den_graph_prepare")
    c1, f1, t1 = tuning(g1)
    assert f1 > 0.0 and t1 > 0.0  # really timed: nothing was cached
    key = "%016x:%s:k6" % (int(lib.tc_den_graph_hash(g1)), torch.cuda.get_device_name(0))  # (hash : device : kernel generation)
    table = json.load(open(path))
    assert table[key]["two_sequence_kernel"] == c1 and table[key]["fused_ms"] == pytest.approx(f1, rel=1e-6)
    g2 = make()
    check(lib.tc_den_graph_prepare(g2, 0), "tc_den_graph_prepare")
    assert tuning(g2) == (c1, 0.0, 0.0)  # the cached choice, no timing launches
    # a choice shipped through tc_tuning_cache_put wins over nothing being cached for a third handle of another process
    assert lib.tc_tuning_cache_put(int(lib.tc_den_graph_hash(g1)), torch.cuda.get_device_name(0).encode(), 1 - c1, 1.0, 1.0) == 0
    g3 = make()
    check(lib.tc_den_graph_prepare(g3, 0), "tc_den_graph_prepare")
    assert tuning(g3)[0] == 1 - c1
    for h in (g1, g2, g3):
        lib.tc_den_graph_free(h)


# ---- positions a tied layout leaves unused (schedule_owner.cpp: phantom positions carry the pdf of their lane) -----------
@pytest.mark.parametrize("H,deg,P,S,T", [(9000, 6, 1500, 5, 9), (13000, 5, 2928, 130, 4), (5000, 8, 40, 3, 8), (18000, 4, 2000, 2, 6)])
def test_unused_positions_add_nothing(oracle, kernel_family, H, deg, P, S, T):
    """A graph whose last plane of 4096 positions is mostly empty (3288 / 3384 / 3192 / 2480 unused positions; 12, 16 and 8 states
    per thread and the plane-wise kernel; one batch above 128 sequences for the fused form; 40 pdfs: fewer than lanes) gives the
    oracle's derivative, and bit for bit the derivative it gave when the unused positions all pointed at pdf 0: they add zero
    wherever they point -- what changed is that 64 lanes no longer queue on one LDS address."""
    fst = synth.random_den_fst(H, deg, P, seed=H + 5)
    y = synth.random_nnet_output(S, T, P, seed=H + 6)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 1 and out["status"] == 0
    ref_lp, ref = oracle_den(oracle, fst, y, S, T, 0.1)
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "unused positions %d" % H)
    kernel_family("phantom_pdf0")
    old = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert old["logprob"] == out["logprob"]
    assert np.array_equal(old["deriv"], out["deriv"])


@pytest.mark.parametrize("form", ["default", "force_mitm", "no_phase_split"])
@pytest.mark.parametrize("cfg,S,T", [("C2", 2, 150), ("R1", 2, 100), ("R3", 2, 60)])
def test_derivative_elementwise_against_float64(kernel_family, cfg, S, T, form):
    """BASELINE.json's "within 1e-4 relative", read element-wise and against the float64 formulation (oracle/independent_f64.py:
    log-semiring, no scaling) instead of the Kaldi-style float32 oracle, whose own distance from it is 2-4e-6 here: on N(0, 1)
    outputs every entry above 1e-4 of the denominator's occupation matrix is within 1e-4 relative and every entry above 1e-3 within
    2e-5 (measured, all three forms, both leaky coefficients: at most 5.9e-5 / 1.02e-5, profiles/r05_gamma_accuracy.txt) -- on the
    metric's graph, on a phone-LM-structured graph and on one that reaches the tied kernels through state splitting; two CUs per
    sequence (the default at this batch, in both of its forms) and the fused kernel."""
    if form != "default":
        kernel_family(form)
    for leaky in (0.1, 1e-5):
        fst, y, lp, ref = float64_truth(cfg, S, T, leaky)
        out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
        assert out["status"] == 0 and abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
        got = np.asarray(out["deriv"], np.float64)
        for floor, tol in ((1e-4, 1e-4), (1e-3, 2e-5)):
            m = ref > floor
            assert m.sum() > 1000
            worst = float((np.abs(got[m] - ref[m]) / ref[m]).max())
            assert worst <= tol, (cfg, form, leaky, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))


@pytest.mark.parametrize("cfg,S,T,flags", [("R4", 2, 40, ()), ("R4", 2, 40, ("no_phase_split",)), ("R4", 2, 40, ("no_planes",)),
                                           ("C2", 2, 60, ("force_general",)), ("C2", 2, 60, ("force_general", "old_general")),
                                           ("C2", 2, 60, ("force_streamed",)), ("C5", 2, 60, ()), ("C5", 2, 60, ("no_phase_split",))])
def test_derivative_elementwise_against_float64_other_kernels(kernel_family, cfg, S, T, flags):
    """The same reading for the other kernel families: the plane-wise kernel in both forms and the streamed path on the 24000-state
    graph, the two general on-chip kernels and the streamed path on the metric's graph, three planes of pdfs (C5).  Measured: at most
    5.9e-5 on entries above 1e-4 and 2.7e-5 on entries above 1e-3 (profiles/r05_gamma_accuracy.txt, last block)."""
    for f in flags:
        kernel_family(f)
    for leaky in (0.1, 1e-5):
        fst, y, lp, ref = float64_truth(cfg, S, T, leaky)
        out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
        assert out["status"] == 0 and abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
        got = np.asarray(out["deriv"], np.float64)
        for floor, tol in ((1e-4, 1e-4), (1e-3, 5e-5)):
            m = ref > floor
            assert m.sum() > 1000
            worst = float((np.abs(got[m] - ref[m]) / ref[m]).max())
            assert worst <= tol, (cfg, flags, leaky, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))
