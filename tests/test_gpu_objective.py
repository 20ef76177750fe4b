"""The full objective through the C ABI (``tc_chain_objf_and_deriv``: numerator + denominator + finalisation) against the CPU
oracle on the same seeded inputs: BASELINE.json configs[0] and configs[1] at full size, subsets of the larger ones, the edge cases
the reference's tests cover (zero outputs, weight 0.5, ragged shapes, non-zero final weights: ``src/chain-supervision-test.hpp:239-341``),
soft numerical failure, peaky outputs, the per-device supervision pool.  Replaces ``src/my_lib_chain.cpp:104-136``.
Tolerance: ``north_star``'s "within 1e-4 relative" -- REL = 1e-4 (scalars relative to the oracle's value; matrices: max abs
difference relative to the oracle's max abs entry, floor = the supervision weight)."""
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


def test_config1_full_objective(oracle):
    """BASELINE.json configs[0]: batch 16, 50 frames, 200 pdf-ids, random 3-state left-to-right den.fst."""
    c = synth.CONFIGS["C1"]
    fst = synth.config_den_fst("C1")
    for leaky in (1e-5, 0.2):  # the two values the reference's test draws (chain-supervision-test.hpp:253-255)
        check_full(oracle, fst, c["S"], c["T"], l2=5e-5, leaky=leaky)


def test_supervision_weight_half(oracle):
    """supervision.weight = 0.5 as in my_lib_chain.cpp:198-199."""
    check_full(oracle, synth.config_den_fst("C1"), 5, 17, l2=1e-3, leaky=0.1, weight=0.5)


def test_zero_nnet_output(oracle):
    """The all-zero nnet output the reference tests with p = 1/4 (chain-supervision-test.hpp:397-399)."""
    check_full(oracle, synth.random_den_fst(96, 5, 40, seed=9), 4, 23, l2=0.0, leaky=1e-5, zero=True)


@pytest.mark.parametrize("H,deg,P,S,T", [(1, 1, 1, 1, 2), (7, 3, 5, 1, 1), (130, 4, 77, 3, 9), (1000, 6, 300, 5, 31)])
def test_ragged_small_shapes(oracle, H, deg, P, S, T):
    """Odd sizes: P and H not multiples of 4 (scalar row path), one state, one frame, one sequence."""
    check_full(oracle, synth.random_den_fst(H, deg, P, seed=H), S, T, l2=1e-4, leaky=0.05, paths=2)


@pytest.mark.parametrize("force", ["force_streamed", "force_streamed,slab_wide", "force_general"])
@pytest.mark.parametrize("H,deg,P,S,T", [(1, 1, 1, 1, 2), (7, 3, 5, 1, 1), (130, 4, 77, 3, 9), (70, 3, 65, 65, 3)])
def test_ragged_small_shapes_other_kernels(oracle, kernel_family, force, H, deg, P, S, T):
    """The same odd sizes through the streamed kernels (sequence counts that are not a multiple of a slab's 16 or
    32 sequences included) and through the general on-chip kernel."""
    for key in force.split(","):
        kernel_family(key)
    check_full(oracle, synth.random_den_fst(H, deg, P, seed=H), S, T, l2=1e-4, leaky=0.05, paths=2)


def test_unaligned_row_stride(oracle):
    """Row stride > num_pdfs and not a multiple of 4: the C ABI takes (rows, cols, row_stride)
    like common::make_matrix (src/common.hpp:109-117)."""
    check_full(oracle, synth.random_den_fst(64, 4, 32, seed=3), 3, 8, l2=1e-4, leaky=0.1, row_pad=3)


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
    check_full(oracle, fst, 6, 40, l2=c["l2"], leaky=c["leaky"])


def test_config5_large_vocab_subset(oracle):
    """Config 5 graph (P=10240, H=8192, A=61440): alpha' does not fit LDS next to gamma -> the
    kernel variant that reads alpha' from the history."""
    c = synth.CONFIGS["C5"]
    fst = synth.config_den_fst("C5")
    assert len(fst.src) == 61440
    from torchain_amd import io
    assert io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"] == 1  # tight tied layout
    check_full(oracle, fst, 3, 20, l2=c["l2"], leaky=c["leaky"])


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


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_config2_full_size_full_objective(oracle, kernel_family, form):
    """configs[1]: CHiME5-like den graph (H=8192, A=65536, P=4096), batch 64 x 150 frames, objf / l2 / weight /
    derivative / xent derivative vs the oracle -- in the two-CU form a batch of 64 takes by default and in the
    fused kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    c = synth.CONFIGS["C2"]
    fst = synth.config_den_fst("C2")
    S, T, P = c["S"], c["T"], c["P"]
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=9, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, P, seed=1236)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, c["l2"], c["leaky"], want_xent=True)
    out = hip_chain(fst, sup, y, l2=c["l2"], leaky=c["leaky"], xent=True)
    res = out["results"]
    assert abs(res[0] - ref["objf"]) <= REL * abs(ref["objf"]), (res, ref["results"])
    assert abs(res[1] - ref["l2_term"]) <= REL * abs(ref["l2_term"])
    assert res[2] == ref["weight"] == S * T
    assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL
    assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL


@pytest.mark.parametrize("scale", [5.0, 10.0])
def test_full_objective_on_peaky_outputs(oracle, scale):
    """objf, l2_term and the whole derivative (numerator + denominator + l2) on y ~ N(0, scale^2), T = 150, against the
    float64 log-semiring formulation; the distance to the Kaldi-style fp32 oracle is bounded by the oracle's own
    distance to that truth (test_gpu_peaky.py's rule: at scale 10 the oracle drifts by 3e-4, the HIP path does not)."""
    from oracle import independent_f64 as ind

    fst = synth.config_den_fst("C2")
    S, T, P = 2, 150, fst.num_pdfs
    l2, leaky = 5e-5, 0.1
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 3, seed=9, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, P, seed=21, scale=scale)
    assert np.abs(y).max() < 30.0 * (scale / 5.0)  # (scale 5: inside the exp clamp; scale 10: a few rows beyond)
    truth = ind.chain_objf_and_deriv(fst, g.initial_probs(), sup, np.clip(y, -30.0, 30.0), l2, leaky)
    truth_deriv = truth["deriv"] + sup.weight * l2 * (np.clip(y, -30, 30) - y)  # the l2 term sees the unclamped y
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky)
    out = hip_chain(fst, sup, y, l2=l2, leaky=leaky)
    res = out["results"]
    truth_l2 = -0.5 * sup.weight * l2 * float((y.astype(np.float64) ** 2).sum())
    assert abs(res[0] - truth["objf"]) <= REL * abs(truth["objf"])
    assert abs(res[1] - truth_l2) <= REL * abs(truth_l2)
    assert res[2] == truth["weight"] == S * T
    assert np.abs(out["deriv"] - truth_deriv).max() <= 2e-5       # absolute: posteriors live in [0, 1]
    assert np.abs(out["deriv"] - ref["deriv"]).max() <= np.abs(ref["deriv"] - truth_deriv).max() + 2e-5
    assert abs(res[0] - ref["objf"]) <= REL * abs(ref["objf"])


def test_native_self_test_entry():
    """tc_self_test: the C-only counterpart of the reference's my_lib_test_chain (src/my_lib_chain.cpp:138-213) --
    weight = w S T, objf <= 0 for a numerator inside the denominator, derivative rows sum to 0, finite differences."""
    import ctypes as C

    from torchain_amd._lib import lib

    rep = (C.c_float * 6)()
    rc = lib.tc_self_test(0, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.cast(rep, C.c_void_p))
    objf, weight, worst_row, predicted, observed, l2_term = list(rep)
    assert rc == 0, (rc, list(rep))
    assert weight == 0.5 * 3 * 9 and objf < 0 and l2_term == 0.0
    assert worst_row <= 1e-4 and abs(observed - predicted) <= 0.1 * abs(predicted) + 1e-4 and predicted != 0.0


def test_objective_without_derivatives_through_the_plain_entry_point(oracle):
    """``tc_chain_objf_and_deriv`` with ``nnet_output_deriv == NULL`` ([K] ``ComputeChainObjfAndDeriv``'s own optional argument; the
    evaluation step of ``tc_chain_step`` is the same computation): results equal to the call with derivatives, on a small and a
    plane-wise graph, in the fused and the two-workgroup forms the derivative call takes."""
    for fst, S, T in ((synth.random_den_fst(300, 5, 90, seed=3), 6, 11), (synth.random_den_fst(17000, 3, 500, seed=4), 3, 7)):
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=8, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=9)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1)
        full = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1)
        none = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, want_deriv=False, graph=full["graph"])
        assert none["deriv"] is None
        np.testing.assert_allclose(none["results"], full["results"], rtol=1e-6)
        assert abs(none["results"][0] - ref["objf"]) <= REL * abs(ref["objf"]) and none["results"][2] == ref["weight"]
