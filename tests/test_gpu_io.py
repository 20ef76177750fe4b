"""The data formats either side of the path, on the GPU: a denominator graph read from an OpenFst ``den.fst`` file
(``src/my_lib_example.cpp:129-134``) and ``chain_loss`` on minibatches from the chain-egs reader (``src/my_lib_example*.cpp``)."""
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


def test_denominator_graph_from_den_fst_file(oracle, tmp_path):
    """The reference's own entry: io.DenominatorGraph(path, n_pdf) (torchain/io.py:51-54 ->
    src/my_lib_example.cpp:129-134) on a den.fst written to disk (with symbol tables), then the full objective
    on the GPU against the oracle built from the in-memory arrays."""
    from torchain_amd import io
    from fixtures import write_openfst_vector

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


def test_chain_loss_on_minibatches_from_the_egs_reader(oracle, tmp_path):
    """The recipe's loop (example/chime5/train_faster.py:79-129): io.RandExample(scp, seed, batchsize) yields
    ((mfcc, ivector), supervision); chain_loss on a network output of the supervision's shape.  The merged
    supervision comes out of the Kaldi-free egs reader and goes through tc_supervision_create; objective and
    derivative against the oracle on the same merged FST."""
    import kaldi_egs_writer as kw
    from fixtures import make_example
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
