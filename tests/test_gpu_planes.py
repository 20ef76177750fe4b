"""The plane-wise on-chip kernel (den_tied_planes.hip): tied graphs of 16385..28672 positions -- the size class of the den.fst
the reference's recipe loads (``example/chime5/train_faster.py:91`` -> ``src/my_lib_example.cpp:129-134``) -- against the oracle at size
(R4 at 64 x 150 and 256 x 30), both forms (one and two workgroups per sequence), accumulate / forward-only, peaky, fuzz, through
``chain_loss``.  Long utterances: tests/test_gpu_long_utterances.py.  REL = 1e-4."""
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


# ---- the plane-wise on-chip kernel (den_tied_planes.hip): tied graphs of 16385..28672 positions --------------------------
@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_small_batches(oracle, kernel_family, form):
    """5, 6 and 7 planes, few sequences and frames, against the full objective's oracle (numerator included) -- in the form
    batches of at most half the CUs take by default (two workgroups per sequence meeting in the middle) and in the fused kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    for H, deg, P, S, T in ((17000, 3, 900, 2, 7), (24000, 4, 2000, 3, 5), (28000, 5, 2928, 2, 6)):
        fst = synth.random_den_fst(H, deg, P, seed=H)
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=H + 1, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, P, seed=H + 2)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True)
        st = out["graph"].stats()
        assert st["tied"] == 1 and st["lds_bytes"] > 100 * 1024, st
        res = out["results"]
        assert abs(res[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T), (H, res, ref["results"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, H
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL, H


@pytest.mark.parametrize("S,T,form", [(64, 150, "two_cu"), (64, 150, "fused"), (256, 30, "fused"), (128, 31, "two_cu")])
def test_plane_wise_kernel_at_size(oracle, kernel_family, S, T, form):
    """R4 on its default path: on chip (tied == 1), log-prob, derivative matrix-wise and element-wise, row sums.  Batches of
    up to 128 sequences take the two-workgroup form (an odd T: the roles' halves differ), 256 the fused kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    compare_at_size(oracle, "R4", S, T, seed=511, expect_tied=1)


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_accumulate_and_no_deriv(oracle, kernel_family, form):
    """[K] Backward(deriv_weight, &deriv) adds into deriv; the forward-only call gives the same log-prob."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(20000, 3, 700, seed=31)
    S, T = 3, 9
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=41)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.05, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert out["graph"].stats()["tied"] == 1 and out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"] - 0.5, ref["deriv"]) <= REL
    out2 = hip_den(fst, y, S, leaky=0.05, want_deriv=False)
    assert abs(out2["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_is_bitwise_reproducible_and_slices(kernel_family, form):
    """Sequences never interact: a 5-sequence call's rows equal the rows of the same sequences in an 8-sequence call."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("R4")
    P = synth.CONFIGS["R4"]["P"]
    S, T = 8, 12
    y = synth.random_nnet_output(S, T, P, seed=43)
    g = io.DenominatorGraph(fst, P)
    a = hip_den(fst, y, S, leaky=0.1, graph=g)
    b = hip_den(fst, y, S, leaky=0.1, graph=g)
    assert a["logprob"] == b["logprob"] and np.array_equal(a["deriv"], b["deriv"])
    sub = np.ascontiguousarray(y.reshape(T, S, P)[:, :5].reshape(T * 5, P))
    c = hip_den(fst, sub, 5, leaky=0.1, graph=g)
    assert np.array_equal(c["deriv"].reshape(T, 5, P), a["deriv"].reshape(T, S, P)[:, :5])


@pytest.mark.parametrize("form", ["two_cu", "fused"])
@pytest.mark.parametrize("scale,leaky", [(10.0, 1e-5), (10.0, 0.1), (20.0, 0.1)])
def test_plane_wise_kernel_peaky_outputs(oracle, kernel_family, scale, leaky, form):
    """y ~ N(0, scale^2), 150 frames, against the float64 log-semiring formulation with tests/test_gpu_peaky.py's bounds
    (absolute 1e-5 on posteriors, element-wise by magnitude class, and no further from the Kaldi-style float32 oracle than
    that oracle is from the truth) -- the plane-wise kernel keeps alpha UN-dashed in its history and beta' in L2."""
    if form == "fused":
        kernel_family("no_phase_split")
    peaky_check(oracle, synth.random_den_fst(17000, 3, 600, seed=77), 1, 150, scale, leaky)


def test_plane_wise_kernel_long_utterance(oracle, kernel_family):
    """400 frames (the frame sums live in LDS: round4(T + 1) floats behind the layout), both forms, log-prob and row sums."""
    fst = synth.random_den_fst(17000, 3, 500, seed=91)
    S, T = 2, 400
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=92)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    for form in ("two_cu", "fused"):
        kernel_family("no_phase_split", 1 if form == "fused" else 0)
        out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
        assert out["status"] == 0 and out["graph"].stats()["tied"] == 1
        assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"]), form
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, form


def planes_fuzz_case(oracle, rng, split):
    """One seeded case of the plane-wise kernel's sweep: random chain-structured graphs, graphs with hub states (secondary rows
    folded per plane, or home rows cut longer), nearly chain-structured graphs the library splits, phone-LM structure; 1 to 5
    sequences (two workgroups per sequence, or the fused kernel), 1 to 12 frames, through the full objective (tests/test_gpu_fuzz.py's
    bounds).  ``split``: graphs of 28673..40960 positions (gather source in LDS a half at a time), else 16385..28672."""
    from torchain_amd._lib import lib
    kind = str(rng.choice(["tied", "hubs", "nearly", "phone_lm"]))
    seed = int(rng.integers(0, 10000))
    P = int(rng.choice([64, 700, 2928, 4096]))
    lo, hi = (28700, 40900) if split else (16500, 28600)
    if kind == "tied":
        fst = synth.random_den_fst(int(rng.integers(lo, hi)), int(rng.integers(2, 7)), P, seed=seed)
    elif kind == "hubs":
        H = int(rng.integers(lo, hi - 1600))
        fst = synth.skewed_tied_den_fst(H, H * int(rng.integers(3, 8)), P, seed=seed, hub_fraction=float(rng.choice([0.002, 0.01])))
    elif kind == "nearly":
        nlo, nhi = (15200, 19000) if split else (9000, 13000)  # (states before the library splits them)
        fst = synth.nearly_tied_den_fst(int(rng.integers(nlo, nhi)), int(rng.integers(3, 6)), P, seed=seed, fraction=float(rng.uniform(0.3, 0.8)))
    else:
        hlo, hhi = (2600, 3300) if split else (1500, 2300)
        fst = synth.phone_lm_den_fst(num_histories=int(rng.integers(hlo, hhi)), branching=int(rng.integers(9, 13)), num_pdfs=max(P, 200), seed=seed)
    S, T = int(rng.integers(1, 6)), int(rng.integers(1, 13))
    leaky, l2 = float(rng.choice([1e-5, 0.05, 0.2])), float(rng.choice([0.0, 1e-4]))
    fused = bool(rng.integers(0, 2))
    lib.tc_debug_set(b"no_phase_split", 1 if fused else 0)
    try:
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=seed + 1, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed + 2, scale=float(rng.choice([1.0, 3.0])))
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky, want_xent=True)
        out = hip_chain(fst, sup, y, l2=l2, leaky=leaky, xent=True)
    finally:
        lib.tc_debug_set(b"no_phase_split", 0)
    st = out["graph"].stats()
    res = out["results"]
    e_obj = abs(res[0] - ref["objf"]) / max(abs(ref["objf"]), 0.05 * S * T)
    e_der, e_x = rel_err(out["deriv"], ref["deriv"], floor=1.0), rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0)
    desc = "%s H=%d A=%d P=%d S=%d T=%d leaky=%g l2=%g fused=%d kernel=%d lds=%d: objf %.1e deriv %.1e xent %.1e" % (
        kind, fst.num_states, len(fst.src), fst.num_pdfs, S, T, leaky, l2, fused, st["tied"], st["lds_bytes"], e_obj, e_der, e_x)
    assert e_obj <= REL and e_der <= REL and e_x <= REL and res[2] == ref["weight"], desc
    # (seven planes fit the LDS only beside at most ~3000 pdfs: 112 KB of gather source + exp(y) + gamma + 16 KB of row sums)
    if kind in ("tied", "phone_lm") and fst.num_pdfs <= 4096 and (16384 < fst.num_states <= 24576 or 28672 < fst.num_states <= 40960):
        assert st["tied"] == 1, desc  # (on chip: the plane-wise kernel)
    return desc


@pytest.mark.parametrize("chunk", range(4))
def test_plane_wise_kernel_fuzz(oracle, chunk):
    """Seeded sweep over the plane-wise kernel's graphs of 5 to 7 planes (planes_fuzz_case)."""
    rng = np.random.default_rng(4200 + chunk)
    for _ in range(6):
        planes_fuzz_case(oracle, rng, split=False)


def test_split_source_kernel_fuzz(oracle):
    """... and of 8 to 10 planes (split gather source); a longer run of the same generator is logged in profiles/r06_fuzz_split.txt."""
    rng = np.random.default_rng(4300)
    for _ in range(4):
        planes_fuzz_case(oracle, rng, split=True)


def test_plane_wise_pairs_with_a_co_tenant():
    """The plane-wise kernel's two-workgroup form pairs workgroups by ticket and hands rows over through flags in global
    memory, like den_tied_mitm.hip (tests/test_gpu_round4.py: test_paired_workgroups_with_a_co_tenant).  With large GEMMs of
    another stream keeping the GPU busy the pairs' workgroups are no longer co-resident by default: no hand-over may fail
    and the results must equal an undisturbed run's bit for bit."""
    import torch
    fst = synth.random_den_fst(17000, 3, 500, seed=61)
    S, T = 96, 40
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=62)
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    quiet = hip_den(fst, y, S, leaky=0.1, graph=graph)
    assert quiet["status"] == 0 and np.isfinite(quiet["logprob"]) and graph.stats()["tied"] == 1
    side = torch.cuda.Stream()
    for rep in range(3):
        busy = occupy_half_the_cus(side, 40)
        out = hip_den(fst, y, S, leaky=0.1, graph=graph)
        side.synchronize()
        assert out["status"] == 0 and out["logprob"] == quiet["logprob"], rep
        assert np.array_equal(out["deriv"], quiet["deriv"]), rep
        del busy


def test_plane_wise_graph_through_chain_loss(oracle):
    """The drop-in call on a graph of the plane-wise class: chain_loss(x (B, C, T), den_graph, supervision) -> loss, results and
    loss.backward() (one tc_chain_step), with the cross-entropy regulariser, against the oracle (torchain/functions.py:62-138)."""
    import torch
    from torchain_amd.functions import chain_loss
    fst = synth.random_den_fst(18000, 3, 400, seed=71)
    B, T, P = 3, 11, fst.num_pdfs
    og = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 2, seed=72, initial_probs=og.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=73)  # rows t * B + b
    ref = oracle.compute_chain_objf_and_deriv(og, sup, y, 5e-5, 0.1, want_xent=True)
    den = io.DenominatorGraph(fst, P)
    assert den.stats()["tied"] == 1 and den.stats()["lds_bytes"] > 100 * 1024
    x = torch.from_numpy(np.ascontiguousarray(y.reshape(T, B, P).transpose(1, 2, 0))).to("cuda:0").requires_grad_(True)
    xe = torch.zeros_like(x).requires_grad_(True)
    loss, results = chain_loss(x, den, io.Supervision.from_synth(sup), l2_regularize=5e-5, leaky_hmm_coefficient=0.1,
                               xent_regularize=0.1, xent_input=xe, kaldi_way=True)
    loss.backward()
    torch.cuda.synchronize()
    got = results.data.numpy()
    assert abs(got[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * B * T) and got[2] == ref["weight"]
    grad = x.grad.cpu().numpy().transpose(2, 0, 1).reshape(T * B, P)
    assert rel_err(-grad, ref["deriv"], floor=1.0) <= REL
    xgrad = xe.grad.cpu().numpy().transpose(2, 0, 1).reshape(T * B, P)
    assert rel_err(-xgrad, 0.1 * ref["xent_deriv"], floor=0.1) <= REL


# ---- 28673..40960 positions: the gather source in LDS a half at a time (round 6) ----------------------------------------------
@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_split_source_kernel_small(oracle, kernel_family, form):
    """Against the oracle at toy batches: log-prob, derivative, row sums; Kaldi's accumulate form; the forward-only call; bitwise
    reproducibility and slice identity (sequences never interact)."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(30000, 3, 900, seed=71)
    S, T = 5, 9
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=72)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    graph = out["graph"]
    assert graph.stats()["tied"] == 1 and out["status"] == 0  # (on chip)
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    assert np.abs(out["deriv"].sum(axis=1, dtype=np.float64) - 1.0).max() <= 1e-4
    again = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=graph)
    assert np.array_equal(out["deriv"], again["deriv"]) and out["logprob"] == again["logprob"]
    acc = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, accumulate=True, init=0.5, graph=graph)
    assert rel_err(acc["deriv"] - 0.5, ref["deriv"]) <= REL
    fwd = hip_den(fst, y, S, leaky=0.1, want_deriv=False, graph=graph)
    assert abs(fwd["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    P = fst.num_pdfs
    part = hip_den(fst, np.ascontiguousarray(y.reshape(T, S, P)[:, 1:4].reshape(T * 3, P)), 3, leaky=0.1, deriv_weight=1.0, graph=graph)
    assert np.array_equal(part["deriv"].reshape(T, 3, P), out["deriv"].reshape(T, S, P)[:, 1:4])


@pytest.mark.parametrize("S,T,form", [(64, 150, "two_cu"), (256, 30, "fused"), (128, 31, "two_cu")])
def test_split_source_kernel_at_size(oracle, kernel_family, S, T, form):
    """X2 (40000 states, 400000 arcs, 4096 pdfs) at size, with the element-wise bounds, one and two workgroups per sequence."""
    if form == "fused":
        kernel_family("no_phase_split")
    compare_at_size(oracle, "X2", S, T, seed=601, expect_tied=1)


def test_split_source_kernel_hub_states_and_full_objective(oracle, kernel_family):
    """A phone-LM-structured graph of 32000 states (in-degrees of hundreds: secondary rows of both halves share their private
    slots) through the full objective, numerator included."""
    fst = synth.phone_lm_den_fst(num_histories=3200, branching=10, seed=11)
    assert fst.num_states > 28672
    out, _ref = check_full(oracle, fst, 4, 12, 5e-5, 0.1, seed=81)
    assert out["graph"].stats()["tied"] == 1


def test_split_source_kernel_peaky_and_float64(oracle, kernel_family):
    """Peaky outputs (T = 150) against the float64 formulation, and the element-wise reading on N(0, 1) outputs: every entry above
    1e-4 within 1e-4 relative (tests/test_gpu_tied.py: test_derivative_elementwise_against_float64)."""
    peaky_check(oracle, synth.random_den_fst(30000, 3, 600, seed=77), 1, 150, 10.0, 0.1)
    for leaky in (0.1,):  # (the float64 formulation of a 400000-arc graph costs the CPU 0.4 s per sequence-frame)
        fst, y, lp, ref = float64_truth("X2", 2, 24, leaky)
        out = hip_den(fst, y, 2, leaky=leaky, deriv_weight=1.0)
        assert out["graph"].stats()["tied"] == 1 and out["status"] == 0 and abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
        got = np.asarray(out["deriv"], np.float64)
        for floor, tol in ((1e-4, 1e-4), (1e-3, 5e-5)):
            m = ref > floor
            assert m.sum() > 1000
            assert float((np.abs(got[m] - ref[m]) / ref[m]).max()) <= tol, (leaky, floor)


def test_split_source_graph_long_utterance(oracle, kernel_family):
    """... and a long utterance (T = 900) on a 10-plane graph."""
    fst = synth.random_den_fst(40000, 3, 4096, seed=91)
    S, T = 2, 900
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=92)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 1 and out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
