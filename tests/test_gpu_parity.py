"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real MI355X.

Tolerance: BASELINE.json's north_star asks for objf / l2_term / d objf/d nnet_output "within 1e-4
relative" of the Kaldi CPU arithmetic; every comparison below uses REL = 1e-4 (scalars: relative to
the oracle's value; matrices: max abs difference relative to the oracle's max abs entry).
"""
import numpy as np
import pytest
import torch

from torchain_amd import io, synth

from helpers import hip_chain, hip_den, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


def _check_full(oracle, fst, S, T, l2, leaky, weight=1.0, seed=5, zero=False, row_pad=0, paths=3):
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, paths, seed=seed + 2, weight=weight, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed, zero=zero)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky, want_xent=True)
    out = hip_chain(fst, sup, y, l2=l2, leaky=leaky, xent=True, row_pad=row_pad)
    res = out["results"]
    assert abs(res[0] - ref["objf"]) <= REL * abs(ref["objf"]), (res, ref["results"])
    assert abs(res[1] - ref["l2_term"]) <= REL * max(abs(ref["l2_term"]), 1e-30), (res, ref["results"])
    assert res[2] == ref["weight"] == weight * S * T  # README.md:12-32 pins weight = w*S*T
    assert rel_err(out["deriv"], ref["deriv"], floor=weight) <= REL
    assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=weight) <= REL
    return out, ref


def test_config1_full_objective(oracle):
    """BASELINE.json configs[0]: batch 16, 50 frames, 200 pdf-ids, random 3-state left-to-right den.fst."""
    c = synth.CONFIGS["C1"]
    fst = synth.config_den_fst("C1")
    for leaky in (1e-5, 0.2):  # the two values the reference's test draws (chain-supervision-test.hpp:253-255)
        _check_full(oracle, fst, c["S"], c["T"], l2=5e-5, leaky=leaky)


def test_supervision_weight_half(oracle):
    """supervision.weight = 0.5 as in my_lib_chain.cpp:198-199."""
    _check_full(oracle, synth.config_den_fst("C1"), 5, 17, l2=1e-3, leaky=0.1, weight=0.5)


def test_zero_nnet_output(oracle):
    """The all-zero nnet output the reference tests with p = 1/4 (chain-supervision-test.hpp:397-399)."""
    _check_full(oracle, synth.random_den_fst(96, 5, 40, seed=9), 4, 23, l2=0.0, leaky=1e-5, zero=True)


@pytest.mark.parametrize("H,deg,P,S,T", [(1, 1, 1, 1, 2), (7, 3, 5, 1, 1), (130, 4, 77, 3, 9), (1000, 6, 300, 5, 31)])
def test_ragged_small_shapes(oracle, H, deg, P, S, T):
    """Odd sizes: P and H not multiples of 4 (scalar row path), one state, one frame, one sequence."""
    _check_full(oracle, synth.random_den_fst(H, deg, P, seed=H), S, T, l2=1e-4, leaky=0.05, paths=2)


@pytest.mark.parametrize("force", ["force_streamed", "force_streamed,slab_wide", "force_general"])
@pytest.mark.parametrize("H,deg,P,S,T", [(1, 1, 1, 1, 2), (7, 3, 5, 1, 1), (130, 4, 77, 3, 9), (70, 3, 65, 65, 3)])
def test_ragged_small_shapes_other_kernels(oracle, kernel_family, force, H, deg, P, S, T):
    """The same odd sizes through the streamed kernels (sequence counts that are not a multiple of a slab's 16 or
    32 sequences included) and through the general on-chip kernel."""
    for key in force.split(","):
        kernel_family(key)
    _check_full(oracle, synth.random_den_fst(H, deg, P, seed=H), S, T, l2=1e-4, leaky=0.05, paths=2)


def test_unaligned_row_stride(oracle):
    """Row stride > num_pdfs and not a multiple of 4: the C ABI takes (rows, cols, row_stride)
    like common::make_matrix (src/common.hpp:109-117)."""
    _check_full(oracle, synth.random_den_fst(64, 4, 32, seed=3), 3, 8, l2=1e-4, leaky=0.1, row_pad=3)


def test_skewed_graph_row_splitting(oracle):
    """Hub states with hundreds of in/out arcs, arbitrary arc->pdf labels, non-final states:
    exercises virtual-row splitting in the schedule."""
    fst = synth.skewed_den_fst(300, 6000, 120, seed=4)
    _check_full(oracle, fst, 4, 15, l2=0.0, leaky=0.1)


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
    _check_full(oracle, fst, 4, 15, l2=1e-4, leaky=0.05)


def test_tied_graph_partial_planes(oracle):
    """State counts that do not fill the 4096-position planes of the tied layout (phantom positions):
    one partly filled plane, and one full plane plus a partly filled one."""
    _check_full(oracle, synth.random_den_fst(1500, 3, 257, seed=21), 2, 7, l2=0.0, leaky=0.1)
    _check_full(oracle, synth.random_den_fst(5000, 3, 300, seed=22), 2, 5, l2=0.0, leaky=0.1)
    # three and four planes: the <JV=4> instantiations (8193..16384 states), small and mid vocabularies
    from torchain_amd import io
    for H, P, seed in ((9000, 5000, 23), (14000, 2000, 24)):
        fst = synth.random_den_fst(H, 3, P, seed=seed)
        assert io.DenominatorGraph(fst, P).stats()["tied"] == 1
        _check_full(oracle, fst, 2, 4, l2=1e-4, leaky=0.1)


def test_tied_tight_layout_mid_vocab(oracle):
    """4097..8192 pdfs with 8192 states: the tied kernel's roomy LDS layout does not fit, the tight one
    (alpha' re-read from the history, exp(y) rewritten in place) does; 6000 pdfs with fewer states fits
    the roomy layout of the same <JV=2, PV=2> instantiation."""
    from torchain_amd import io
    fst = synth.random_den_fst(8192, 3, 6000, seed=41)
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1
    _check_full(oracle, fst, 2, 9, l2=1e-4, leaky=0.1)
    _check_full(oracle, synth.random_den_fst(3000, 4, 6000, seed=42), 3, 7, l2=0.0, leaky=0.05)


def test_nearly_tied_graph_state_splitting(oracle):
    """States entered through several pdfs are split into one copy per pdf (exact) so that the graph stays
    on the tied kernel; compared with the oracle run on the ORIGINAL graph."""
    from torchain_amd import io
    fst = synth.nearly_tied_den_fst(700, 5, 150, seed=15, fraction=0.05)
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1
    _check_full(oracle, fst, 3, 12, l2=1e-4, leaky=0.1)
    _check_full(oracle, synth.nearly_tied_den_fst(64, 4, 20, seed=6, fraction=0.3), 2, 9, l2=0.0, leaky=1e-5)


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
def test_streamed_path_one_frame_of_exp_at_a_time(oracle, kernel_family, width):
    """Beyond 1 GB of transposed exp(y) the streamed path keeps one frame of it and recomputes it in the backward pass;
    forced here at test sizes (tied and general graphs, Kaldi's accumulate form)."""
    kernel_family(width)
    kernel_family("force_streamed")
    kernel_family("exp_per_frame")
    _check_full(oracle, synth.random_den_fst(300, 5, 100, seed=32), 3, 11, l2=1e-4, leaky=0.05)
    _check_full(oracle, synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 2, 9, l2=0.0, leaky=0.1)
    fst2 = synth.skewed_den_fst(300, 6000, 120, seed=4)
    _check_full(oracle, fst2, 4, 9, l2=1e-3, leaky=0.1)
    S, T = 3, 8
    y = synth.random_nnet_output(S, T, fst2.num_pdfs, seed=9)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst2), y, S, leaky=0.05, deriv_weight=1.0)
    out = hip_den(fst2, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"]) and out["status"] == 0
    assert rel_err(out["deriv"] - 0.5, ref["deriv"]) <= REL


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
def test_streamed_path_for_graphs_beyond_lds(oracle, kernel_family, width):
    """Graphs the on-chip layouts cannot hold (more than 16384 states here) take the streamed kernel
    (alpha/beta in global memory, slabs of 16 or 32 sequences); the same kernel forced onto small graphs, tied and
    general, must agree with the oracle too, including Kaldi's accumulate form."""
    from torchain_amd import io
    kernel_family(width)
    kernel_family("no_planes")  # (since round 5 a 20000-state tied graph would take the plane-wise on-chip kernel)
    fst = synth.random_den_fst(20000, 3, 700, seed=31)
    g = io.DenominatorGraph(fst, fst.num_pdfs)
    assert g.stats()["tied"] == 2
    _check_full(oracle, fst, 2, 6, l2=1e-4, leaky=0.1)
    kernel_family("force_streamed")
    _check_full(oracle, synth.random_den_fst(300, 5, 100, seed=32), 3, 11, l2=0.0, leaky=1e-5)  # tied streamed kernels
    _check_full(oracle, synth.nearly_tied_den_fst(500, 5, 90, seed=7), 3, 8, l2=1e-4, leaky=0.1)  # ... of a split graph
    _check_full(oracle, synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 2, 9, l2=0.0, leaky=0.05)  # hubs, no-self-loop states
    fst2 = synth.skewed_den_fst(300, 6000, 120, seed=4)
    _check_full(oracle, fst2, 4, 9, l2=1e-3, leaky=0.1)
    S, T = 3, 8
    og = oracle.DenGraph(fst2)
    y = synth.random_nnet_output(S, T, fst2.num_pdfs, seed=9)
    ref = oracle.den_forward_backward(og, y, S, leaky=0.05, deriv_weight=1.0)
    out = hip_den(fst2, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert io.DenominatorGraph(fst2, fst2.num_pdfs).stats()["tied"] == 2
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"]) and out["status"] == 0
    assert rel_err(out["deriv"] - 0.5, ref["deriv"]) <= REL
    out3 = hip_den(fst2, y, S, leaky=0.05, want_deriv=False)
    assert abs(out3["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])


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


def test_numerical_failure_is_soft(oracle):
    """[K] NaN/inf objf -> derivs zeroed (then the l2 derivative is added), objf = -10*weight."""
    fst = synth.random_den_fst(40, 4, 20, seed=6)
    S, T = 2, 6
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=1, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=1)
    y[3, 5] = np.nan
    out = hip_chain(fst, sup, y, l2=0.0, leaky=1e-5, xent=True)
    assert out["results"][0] == -10.0 * S * T
    assert out["results"][2] == S * T
    assert np.all(out["deriv"][np.isfinite(y)] == 0) and np.all(out["xent_deriv"] == 0)


def test_medium_config2_shape_subset(oracle):
    """CHiME5-like graph of config 2/3 (H=8192, A=65536, P=4096) at a batch the oracle finishes in
    seconds; full objective."""
    c = synth.CONFIGS["C2"]
    fst = synth.config_den_fst("C2")
    _check_full(oracle, fst, 6, 40, l2=c["l2"], leaky=c["leaky"])


def test_config5_large_vocab_subset(oracle):
    """Config 5 graph (P=10240, H=8192, A=61440): alpha' does not fit LDS next to gamma -> the
    kernel variant that reads alpha' from the history."""
    c = synth.CONFIGS["C5"]
    fst = synth.config_den_fst("C5")
    assert len(fst.src) == 61440
    from torchain_amd import io
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1  # tight tied layout
    _check_full(oracle, fst, 3, 20, l2=c["l2"], leaky=c["leaky"])


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


def test_autograd_wrapper_matches_reference_semantics(oracle):
    """chain_loss(): (B, C, T) input, loss = -objf/weight, backward = -(deriv) ignoring grad_output,
    xent grad scaled by xent_regularize (torchain/functions.py:62-138)."""
    from torchain_amd import io
    from torchain_amd.functions import chain_loss

    fst = synth.random_den_fst(120, 5, 64, seed=21)
    B, T, P = 4, 12, 64
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 3, seed=3, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=8)  # rows t*B + b
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 5e-5, 0.1, want_xent=True)

    den = io.DenominatorGraph(fst, P)
    hsup = io.Supervision.from_synth(sup)
    x = torch.from_numpy(y.reshape(T, B, P).transpose(1, 2, 0).copy()).cuda().requires_grad_(True)  # (B, C, T)
    xe = torch.randn(B, P, T, device="cuda", requires_grad=True)
    loss, results = chain_loss(x, den, hsup, l2_regularize=5e-5, leaky_hmm_coefficient=0.1, xent_regularize=0.1,
                               xent_input=xe, kaldi_way=True)
    assert loss.is_cuda and loss.shape == (1,)
    assert abs(float(loss) - (-ref["objf"] / ref["weight"])) <= REL * abs(ref["objf"] / ref["weight"])
    (loss * 123.0).backward()  # grad_output must be ignored
    gx = x.grad.permute(2, 0, 1).reshape(T * B, P).cpu().numpy()
    assert rel_err(gx, -ref["deriv"]) <= REL
    gxe = xe.grad.permute(2, 0, 1).reshape(T * B, P).cpu().numpy()
    assert rel_err(gxe, -0.1 * ref["xent_deriv"]) <= REL
    assert "ChainResults(loss=" in repr(results)


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


def test_numerator_beside_the_denominator_changes_nothing(oracle, kernel_family):
    """Small batches leave CUs idle under the denominator: the numerator's recursion then runs on a side stream and
    its posteriors are added once the denominator has written the derivative.  Same bits as one after the other."""
    from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv
    fst = synth.random_den_fst(700, 5, 300, seed=19)
    S, T = 6, 31
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=5, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=6)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    outs = []
    for serial in (0, 1):
        kernel_family("no_num_overlap", serial)
        outs.append(hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True))
    a, b = outs
    assert np.array_equal(a["results"], b["results"])
    assert np.array_equal(a["deriv"], b["deriv"]) and np.array_equal(a["xent_deriv"], b["xent_deriv"])
    assert abs(a["results"][0] - ref["objf"]) <= REL * abs(ref["objf"])
    assert rel_err(a["deriv"], ref["deriv"], floor=1.0) <= REL and rel_err(a["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_gradient_form_is_the_exact_negative(oracle, kernel_family, form):
    """tc_chain_objf_and_grad writes what the reference's backward returns (functions.py:106-115): -deriv and
    -xent_regularize * xent_deriv, bit for bit what negating / scaling tc_chain_objf_and_deriv's outputs gives; the
    three results are unchanged.  Also on the numerical-failure exit (deriv = -w*l2*y, xent_deriv = 0)."""
    from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(300, 5, 120, seed=17)
    S, T, xr = 4, 23, 0.1
    g = oracle.DenGraph(fst)
    sup = io.Supervision.from_synth(synth.random_supervision(fst, S, T, 2, seed=5, initial_probs=g.initial_probs()))
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    for bad in (False, True):
        y = torch.from_numpy(synth.random_nnet_output(S, T, fst.num_pdfs, seed=6, scale=2.0)).cuda()
        if bad:
            y[5, 7] = float("nan")
        outs = []
        for as_grad in (False, True):
            res = ChainResults()
            d = torch.full_like(y, 9.0)
            x = torch.full_like(y, 9.0)
            compute_chain_objf_and_deriv(graph, sup, y, res.data, d, x, 1e-3, 0.05, xr, as_gradients=as_grad)
            outs.append((res.data.clone(), d, x))
        (r0, d0, x0), (r1, d1, x1) = outs
        assert torch.allclose(r0, r1, rtol=0, atol=0, equal_nan=True)  # (l2_term is NaN on the failure exit, as [K]'s)
        if bad:
            assert float(r0[0]) == -10.0 * float(r0[2])
            keep = ~torch.isnan(y)
            assert torch.equal(d1[keep], -d0[keep]) and float(x0.abs().max()) == 0.0 and float(x1.abs().max()) == 0.0
        else:
            assert torch.equal(d1, -d0)
            assert torch.equal(x1, torch.tensor(-xr, dtype=torch.float32) * x0)


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


def test_fused_layout_kernels_match_torch_permute():
    """tc_to2d / tc_from2d against the reference's own layout code (functions.py:118-125: permute(2,0,1)
    .contiguous()); copies, so bit-exact; ragged channel counts and more frames than one tile."""
    from torchain_amd.functions import from2d_hip, to2d, to2d_hip
    gen = torch.Generator(device="cuda").manual_seed(3)
    for B, C, T in ((3, 100, 17), (2, 64, 240), (5, 130, 241), (1, 1, 1), (4, 257, 500)):
        x = torch.randn(B, C, T, device="cuda", generator=gen)
        ref = to2d(x)
        assert torch.equal(to2d_hip(x), ref)
        back = from2d_hip(ref, (B, C, T), -0.5)
        assert torch.equal(back, -0.5 * x)


def test_chain_loss_3d_input_fused_path_matches_2d_path(oracle):
    """chain_loss on a (B, C, T) tensor (fused layout passes) gives the loss and the input / xent-input
    gradients of the reference composition to2d -> 2-D loss -> autograd's inverse permute, bit for bit."""
    from torchain_amd import io
    from torchain_amd.functions import _ChainLoss, ChainResults, chain_loss, to2d
    fst = synth.random_den_fst(200, 5, 90, seed=13)
    B, T, P = 4, 23, 90
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 3, seed=4, initial_probs=g.initial_probs())
    den, hsup = io.DenominatorGraph(fst, P), io.Supervision.from_synth(sup)
    gen = torch.Generator(device="cuda").manual_seed(5)
    for kaldi_way in (True, False):
        x = torch.randn(B, P, T, device="cuda", generator=gen).requires_grad_(True)
        xe = torch.randn(B, P, T, device="cuda", generator=gen).requires_grad_(True)
        loss, res = chain_loss(x, den, hsup, l2_regularize=1e-4, leaky_hmm_coefficient=0.05, xent_regularize=0.1,
                               xent_input=xe, kaldi_way=kaldi_way)
        loss.backward()
        x2 = x.detach().clone().requires_grad_(True)
        xe2 = xe.detach().clone().requires_grad_(True)
        res2 = ChainResults()
        loss2 = _ChainLoss.apply(to2d(x2), to2d(xe2), res2, den, hsup, 1e-4, 0.05, 0.1, kaldi_way)
        loss2.backward()
        assert torch.equal(res.data, res2.data) and torch.equal(loss, loss2)
        assert torch.equal(x.grad, x2.grad) and torch.equal(xe.grad, xe2.grad)


def test_denominator_graph_from_den_fst_file(oracle, tmp_path):
    """The reference's own entry: io.DenominatorGraph(path, n_pdf) (torchain/io.py:51-54 ->
    src/my_lib_example.cpp:129-134) on a den.fst written to disk (with symbol tables), then the full objective
    on the GPU against the oracle built from the in-memory arrays."""
    from torchain_amd import io
    from test_abi import write_openfst_vector

    fst = synth.nearly_tied_den_fst(900, 5, 200, seed=17, fraction=0.05)
    path = str(tmp_path / "den.fst")
    write_openfst_vector(path, fst, with_symbols=True)
    graph = io.DenominatorGraph(path, fst.num_pdfs)
    assert graph.num_states == fst.num_states and graph.num_arcs == len(fst.src)
    S, T = 4, 25
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=18, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=19)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True, graph=graph)
    assert abs(out["results"][0] - ref["objf"]) <= REL * abs(ref["objf"])
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL


def test_supervision_with_nonzero_final_weights(oracle):
    """The f_i - f_0 re-weighting branch of the per-sequence split (csrc/supervision.cpp): boundary states with
    distinct non-zero final weights folded into their arc copies, as after [K] AddWeightToSupervisionFst +
    AppendSupervision.  Objective, posteriors and the xent side output against the oracle on the MERGED FST."""
    fst = synth.random_den_fst(150, 5, 60, seed=51)
    S, T = 5, 14
    g = oracle.DenGraph(fst)
    for weight in (1.0, 0.5):
        sup = synth.random_supervision(fst, S, T, 3, seed=52, weight=weight, initial_probs=g.initial_probs(),
                                       final_weights=True)
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=53)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True)
        assert abs(out["results"][0] - ref["objf"]) <= REL * abs(ref["objf"])
        assert rel_err(out["deriv"], ref["deriv"], floor=weight) <= REL
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=weight) <= REL


def test_fresh_supervision_every_step_allocates_nothing_after_warmup(oracle):
    """A training loop makes a new Supervision per minibatch (reference: io.py:20-31 + the egs iterators).  The
    tables live in slots of a per-device pool with pinned staging: after a few steps the pool's device
    allocation count stops moving although every step creates, uploads, uses and drops a supervision -- and
    the results stay right while slots are being recycled under in-flight kernels."""
    from torchain_amd import io
    from torchain_amd._lib import lib
    from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv

    fst = synth.random_den_fst(200, 5, 80, seed=61)
    S, T, P = 6, 20, 80
    g = oracle.DenGraph(fst)
    graph = io.DenominatorGraph(fst, P)
    sups = [synth.random_supervision(fst, S, T, 3, seed=62 + i, initial_probs=g.initial_probs()) for i in range(4)]
    y_np = synth.random_nnet_output(S, T, P, seed=63)
    refs = [oracle.compute_chain_objf_and_deriv(g, sp, y_np, 0.0, 0.1)["objf"] for sp in sups]
    y = torch.from_numpy(y_np).cuda()
    deriv = torch.empty_like(y)
    side = torch.cuda.Stream()
    allocs = []
    objfs = []
    for step in range(50):
        h = io.Supervision.from_synth(sups[step % 4])
        res = ChainResults()
        if step % 5 == 4:  # every few steps from another stream: prepare must order that stream behind the upload
            with torch.cuda.stream(side):
                compute_chain_objf_and_deriv(graph, h, y, res.data, deriv, None, 0.0, 0.1, 0.0)
            side.synchronize()
        else:
            compute_chain_objf_and_deriv(graph, h, y, res.data, deriv, None, 0.0, 0.1, 0.0)
        objfs.append(float(res.data[0]))
        del h
        allocs.append(lib.tc_debug_counter(b"pool_device_allocs"))
    assert allocs[-1] == allocs[9], allocs           # nothing allocated after the first ten steps
    assert lib.tc_debug_counter(b"pool_reuses") >= 40
    for step, v in enumerate(objfs):
        assert abs(v - refs[step % 4]) <= REL * abs(refs[step % 4])


def test_xent_objective_value_and_rccl_branch(oracle):
    """(a) ChainResults.xent_objf = sum(xent_output * w * numerator posteriors), the cross-entropy objective
    Kaldi's chain trainer logs (a TODO in the reference, torchain/functions.py:88-89), against the oracle's
    xent derivative; (b) chain_loss_data_parallel through the REAL RCCL path with a process group of one rank
    (the recipe's per-device loss, example/chime5/parallel_train.py:59-75: results summed over devices,
    loss = -sum(objf) / sum(weight))."""
    import os
    import torch.distributed as dist
    from torchain_amd import io, parallel
    from torchain_amd.functions import chain_loss

    fst = synth.random_den_fst(120, 5, 64, seed=71)
    B, T, P = 4, 15, 64
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 3, seed=72, weight=0.5, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=73)
    xe = torch.log_softmax(torch.from_numpy(synth.random_nnet_output(B, T, P, seed=74)), dim=1).numpy()
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    want_xent_objf = float((xe.astype(np.float64) * ref["xent_deriv"].astype(np.float64)).sum())
    den, hsup = io.DenominatorGraph(fst, P), io.Supervision.from_synth(sup)
    to3d = lambda a: torch.from_numpy(a.reshape(T, B, P).transpose(1, 2, 0).copy()).cuda()
    loss, res = chain_loss(to3d(y).requires_grad_(True), den, hsup, 1e-4, 0.1, 0.1, to3d(xe).requires_grad_(True),
                           kaldi_way=True)
    assert abs(res.xent_objf - want_xent_objf) <= REL * abs(want_xent_objf)
    assert abs(res.xent_loss - (-want_xent_objf / ref["weight"])) <= REL * abs(want_xent_objf / ref["weight"])
    _, res0 = chain_loss(to3d(y), den, hsup, 1e-4, 0.1)
    assert res0.xent_objf is None and res0.xent_loss is None

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        loss2, res2 = parallel.chain_loss_data_parallel(to3d(y).requires_grad_(True), den, hsup, 1e-4, 0.1, 0.1,
                                                        to3d(xe).requires_grad_(True), kaldi_way=True, even_if_alone=True)
        assert dist.get_backend() == "nccl"
        assert torch.equal(res2.data, res.data) and float(loss2) == float(loss)
        assert abs(res2.xent_objf - res.xent_objf) <= 1e-9 * abs(res.xent_objf)
    finally:
        dist.destroy_process_group()


def test_chain_loss_on_minibatches_from_the_egs_reader(oracle, tmp_path):
    """The recipe's loop (example/chime5/train_faster.py:79-129): io.RandExample(scp, seed, batchsize) yields
    ((mfcc, ivector), supervision); chain_loss on a network output of the supervision's shape.  The merged
    supervision comes out of the Kaldi-free egs reader and goes through tc_supervision_create; objective and
    derivative against the oracle on the same merged FST."""
    import kaldi_egs_writer as kw
    from test_egs import make_example
    from torchain_amd import egs, io
    from torchain_amd.functions import chain_loss

    fst = synth.random_den_fst(80, 4, 40, seed=81)
    keyed = [("utt%02d" % i, make_example(fst, L, seed=300 + i, final_weights=True)) for i, L in enumerate([7, 7, 7, 10, 10])]
    ark, scp = str(tmp_path / "egs.ark"), str(tmp_path / "egs.scp")
    kw.write_ark(ark, keyed, scp_path=scp, matrix_kind="CM")
    g = oracle.DenGraph(fst)
    den = io.DenominatorGraph(fst, 40)
    rd = io.RandExample(scp, seed=1, batchsize=3)
    n = 0
    for (mfcc, ivec), sup in rd:
        B, T, P = sup.shape
        merged = rd._cur["outputs"][0]["supervision"]
        y = synth.random_nnet_output(B, T, P, seed=90 + n)
        ref = oracle.compute_chain_objf_and_deriv(g, merged, y, 1e-4, 0.1)
        x = torch.from_numpy(y.reshape(T, B, P).transpose(1, 2, 0).copy()).cuda().requires_grad_(True)
        loss, res = chain_loss(x, den, sup, l2_regularize=1e-4, leaky_hmm_coefficient=0.1)
        loss.backward()
        assert abs(float(res.data[0]) - ref["objf"]) <= REL * abs(ref["objf"])
        gx = x.grad.permute(2, 0, 1).reshape(T * B, P).cpu().numpy()
        assert rel_err(gx, -ref["deriv"], floor=1.0) <= REL
        n += 1
    assert n == rd.n_batch == 2
