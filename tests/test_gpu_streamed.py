"""The streamed path (den_slab_kernel.hip): graphs beyond the on-chip layouts (more than 28672 positions, or forced), alpha / beta
in HBM, slabs of 16 or 32 sequences -- against the oracle at size (R4 forced, X2) and through the size-independent properties at
256 x 150.  REL = 1e-4."""
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


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
def test_streamed_path_one_frame_of_exp_at_a_time(oracle, kernel_family, width):
    """Beyond 1 GB of transposed exp(y) the streamed path keeps one frame of it and recomputes it in the backward pass;
    forced here at test sizes (tied and general graphs, Kaldi's accumulate form)."""
    kernel_family(width)
    kernel_family("force_streamed")
    kernel_family("exp_per_frame")
    check_full(oracle, synth.random_den_fst(300, 5, 100, seed=32), 3, 11, l2=1e-4, leaky=0.05)
    check_full(oracle, synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 2, 9, l2=0.0, leaky=0.1)
    fst2 = synth.skewed_den_fst(300, 6000, 120, seed=4)
    check_full(oracle, fst2, 4, 9, l2=1e-3, leaky=0.1)
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
    check_full(oracle, fst, 2, 6, l2=1e-4, leaky=0.1)
    kernel_family("force_streamed")
    check_full(oracle, synth.random_den_fst(300, 5, 100, seed=32), 3, 11, l2=0.0, leaky=1e-5)  # tied streamed kernels
    check_full(oracle, synth.nearly_tied_den_fst(500, 5, 90, seed=7), 3, 8, l2=1e-4, leaky=0.1)  # ... of a split graph
    check_full(oracle, synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 2, 9, l2=0.0, leaky=0.05)  # hubs, no-self-loop states
    fst2 = synth.skewed_den_fst(300, 6000, 120, seed=4)
    check_full(oracle, fst2, 4, 9, l2=1e-3, leaky=0.1)
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


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
@pytest.mark.parametrize("cfg", ["R4", "X2"])
def test_streamed_path_at_size_64x150(oracle, kernel_family, cfg, width):
    kernel_family("force_streamed")
    kernel_family(width)
    compare_at_size(oracle, cfg, 64, 150, seed=501, expect_tied=2)


@pytest.mark.parametrize("cfg,width", [("R4", "slab_narrow"), ("X2", "slab_wide")])
def test_streamed_path_at_size_256x30(oracle, kernel_family, cfg, width):
    """more slabs than XCDs (16 / 8 slabs of 16 / 32 sequences): every XCD walks several slabs one after the other"""
    kernel_family("force_streamed")
    kernel_family(width)
    compare_at_size(oracle, cfg, 256, 30, seed=502, expect_tied=2)


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
