"""Round 5 ("Next round" item 3 and weak #3): the kernels for graphs beyond the 8/12/16-states-per-thread
on-chip layouts compared with the oracle AT SIZE -- R4 (24 000 states, 312 000 arcs, phone-LM structure with in-degrees
to 200: the size class of the reference recipe's own den.fst, example/chime5/train_faster.py:91) and X2 (40 000 states,
400 000 arcs) at 64 x 150 and 256 x 30, both slab widths of the streamed path (many slabs per XCD, the XCD-aware block
decode with more than 8 blocks per XCD, hub bundles with groups_sum, fixed-point L2 gamma atomics over 150 frames), with
the matrix-wise and the element-wise bounds of test_gpu_round4.py, plus the size-independent property sum(gamma) = S*T and
row sums = 1 at 256 x 150 ([K] DenominatorComputation's own check, the reference's src/chain-supervision-test.hpp:388-463).
The oracle runs its sequences in blocks on the host's cores (oracle.den_forward_backward_blocks)."""
import os

import numpy as np
import pytest

from torchain_amd import io, synth

from helpers import hip_den, rel_err
from test_gpu_round4 import elementwise

pytestmark = pytest.mark.gpu
REL = 1e-4


def _oracle_den(oracle, fst, y, S, T, leaky):
    threads = max(1, min(64, os.cpu_count() or 1))
    lp, deriv = oracle.den_forward_backward_blocks(oracle.DenGraph(fst), y, S, T, leaky, block=max(1, S // threads),
                                                   threads=threads, deriv_weight=1.0)
    return lp, deriv


def _compare(oracle, cfg, S, T, seed, expect_tied):
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    y = synth.random_nnet_output(S, T, c["P"], seed=seed)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == expect_tied, out["graph"].stats()
    assert out["status"] == 0
    ref_lp, ref = _oracle_den(oracle, fst, y, S, T, c["leaky"])
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp), (out["logprob"], ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "%s %dx%d" % (cfg, S, T))
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() <= 1e-4, np.abs(rows - 1.0).max()


# ---- the plane-wise on-chip kernel (den_tied_planes.hip): tied graphs of 16385..28672 positions --------------------------
@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_small_batches(oracle, kernel_family, form):
    """5, 6 and 7 planes, few sequences and frames, against the full objective's oracle (numerator included) -- in the form
    batches of at most half the CUs take by default (two workgroups per sequence meeting in the middle) and in the fused kernel."""
    from helpers import hip_chain
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
    _compare(oracle, "R4", S, T, seed=511, expect_tied=1)


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
    from test_gpu_peaky import _check
    if form == "fused":
        kernel_family("no_phase_split")
    _check(oracle, synth.random_den_fst(17000, 3, 600, seed=77), 1, 150, scale, leaky)


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
@pytest.mark.parametrize("cfg", ["R4", "X2"])
def test_streamed_path_at_size_64x150(oracle, kernel_family, cfg, width):
    kernel_family("force_streamed")
    kernel_family(width)
    _compare(oracle, cfg, 64, 150, seed=501, expect_tied=2)


@pytest.mark.parametrize("cfg,width", [("R4", "slab_narrow"), ("X2", "slab_wide")])
def test_streamed_path_at_size_256x30(oracle, kernel_family, cfg, width):
    """more slabs than XCDs (16 / 8 slabs of 16 / 32 sequences): every XCD walks several slabs one after the other"""
    kernel_family("force_streamed")
    kernel_family(width)
    _compare(oracle, cfg, 256, 30, seed=502, expect_tied=2)


@pytest.mark.parametrize("cfg", ["R4", "X2"])
def test_streamed_path_gamma_sums_at_256x150(kernel_family, cfg):
    """[K] BetaGeneralFrameDebug's invariant at the full batch: every frame's posteriors sum to one, so the derivative sums
    to S*T (no oracle needed; 150 frames of fixed-point L2 atomics near their range)."""
    kernel_family("force_streamed")
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    S, T = 256, 150
    y = synth.random_nnet_output(S, T, c["P"], seed=503)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["status"] == 0 and out["graph"].stats()["tied"] == 2
    assert out["deriv"].min() >= 0.0
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() <= 1e-4, np.abs(rows - 1.0).max()
    assert abs(rows.sum() - S * T) <= 1e-5 * S * T
