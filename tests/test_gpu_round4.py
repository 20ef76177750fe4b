"""Round 4 ("Next round" items 4 and 7):

* the per-graph kernel choice is REPRODUCIBLE: a second handle of the same graph takes the first one's measured choice
  from the cache (``io.DenominatorGraph.prepare``) without timing anything and its derivatives are bit-identical; a
  choice fixed by the caller (``tc_den_graph_set_variant``) is honoured, and both kernels agree with the oracle
  (the reference's call is deterministic for fixed inputs: ``src/my_lib_chain.cpp:129-131``);
* ELEMENT-WISE derivative checks at full size (C2, C5, R1, R3; C3 has had one since round 2): entries above 1e-3 within
  1e-4 relative, entries above 1e-4 within 1e-3 -- ``north_star``'s "within 1e-4 relative" read per element where an
  element is large enough to carry four digits in float32 posteriors (DESIGN.md section 1 says where the tied kernels'
  subtraction form of gamma stops: entries of 1e-6 are a few per cent off);
* co-tenancy: the kernels whose workgroups wait for each other (two CUs per sequence meeting in the middle, two
  sequences per workgroup) launched while a long kernel on another stream holds half of the CUs -- no failed
  hand-over, results identical to an undisturbed run.
(Round 4 also tested an opt-in register-row kernel, den_tied_rr.hip; round 5 removed that kernel -- VERDICT: "if it does not become
the base of [the large-graph kernel], delete it" -- and keeps its measurements in profiles/r04_ablations.txt section 3.)
Reference property tests these restate: ``src/chain-supervision-test.hpp:239-341``."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from torchain_amd import io, synth
from torchain_amd._lib import check, lib

from helpers import hip_chain, hip_den, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


def elementwise(got, ref, what, bounds=((1e-3, 1e-4), (1e-4, 1e-3))):
    """entries of |ref| > 1e-3 within 1e-4 relative, entries > 1e-4 within 1e-3 relative"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    for floor, tol in bounds:
        m = np.abs(ref) > floor
        assert m.any(), (what, floor)
        worst = float((np.abs(got[m] - ref[m]) / np.abs(ref[m])).max())
        assert worst <= tol, (what, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))


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

    from test_oracle_golden import GOLDEN, load
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


def _occupy_half_the_cus(stream, millis):
    """A long kernel on ``stream`` that holds half of the CUs: torch kernels sized to the device, far more work than
    the launches under test (a matmul chain of ~`millis` ms on 128 of the 256 CUs' worth of workgroups)."""
    n = 2048
    a = torch.randn(n, n, device="cuda")
    with torch.cuda.stream(stream):
        x = a
        for _ in range(max(1, millis // 2)):
            x = torch.tanh(x @ a * 1e-3)
    return x


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
        busy = _occupy_half_the_cus(side, 40)
        out = hip_den(fst, y, S, leaky=0.1, graph=graph)
        side.synchronize()
        assert out["status"] == 0 and out["logprob"] == quiet["logprob"], (mode, rep)
        assert np.array_equal(out["deriv"], quiet["deriv"]), (mode, rep)
        del busy


@pytest.mark.parametrize("three_d", [True, False])
@pytest.mark.parametrize("kaldi_way", [True, False])
def test_one_call_step_equals_the_multi_call_wrappers(three_d, kaldi_way):
    """``tc_chain_step`` (what ``chain_loss`` calls for CUDA float32 tensors) against ``_ChainLoss`` / ``_ChainLoss3d``
    (four to six library calls): loss, results, xent objective and both gradients, bit for bit."""
    from torchain_amd.functions import ChainResults, _ChainLoss, _ChainLoss3d, chain_loss
    fst = synth.random_den_fst(300, 4, 96, seed=4)
    S, T, P = 5, 11, 96
    den = io.DenominatorGraph(fst, P)
    sup = synth.random_supervision(fst, S, T, 3, seed=2, initial_probs=den.initial_probs())
    hsup = io.Supervision.from_synth(sup)
    y = torch.from_numpy(synth.random_nnet_output(S, T, P, seed=6)).cuda()
    xe = torch.from_numpy(synth.random_nnet_output(S, T, P, seed=7)).cuda()
    if three_d:
        y = y.reshape(T, S, P).permute(1, 2, 0).contiguous()
        xe = xe.reshape(T, S, P).permute(1, 2, 0).contiguous()
    outs = []
    for one_call in (True, False):
        a, b = y.clone().requires_grad_(True), xe.clone().requires_grad_(True)
        if one_call:
            loss, res = chain_loss(a, den, hsup, 1e-4, 0.05, 0.1, b, kaldi_way)
        else:
            res = ChainResults()
            loss = (_ChainLoss3d if three_d else _ChainLoss).apply(a, b, res, den, hsup, 1e-4, 0.05, 0.1, kaldi_way)
        loss.backward()
        outs.append((loss.detach().cpu(), res.data.clone(), res.xent_objf, a.grad.cpu(), b.grad.cpu()))
    (l1, r1, x1, g1, xg1), (l2, r2, x2, g2, xg2) = outs
    assert torch.equal(r1, r2) and torch.equal(g1, g2) and torch.equal(xg1, xg2)
    assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2)) and abs(x1 - x2) <= 1e-6 * abs(x2)
    # without a xent branch, and a 2-D input whose rows are not contiguous
    a = y.clone().requires_grad_(True)
    loss, res = chain_loss(a, den, hsup, 1e-4, 0.05)
    loss.backward()
    res0 = ChainResults()
    b = y.clone().requires_grad_(True)
    (_ChainLoss3d if three_d else _ChainLoss).apply(b, None, res0, den, hsup, 1e-4, 0.05).backward()
    assert torch.equal(res.data, res0.data) and torch.equal(a.grad, b.grad) and res.xent_objf is None
    if not three_d:
        wide = torch.zeros(S * T, P + 8, device="cuda")
        wide[:, :P] = y
        c = wide[:, :P].detach().requires_grad_(True)
        loss, res2 = chain_loss(c, den, hsup, 1e-4, 0.05)
        loss.backward()
        assert torch.equal(res2.data, res.data) and torch.equal(c.grad, a.grad)


# ---- two ranks fed from the native reader ------------------------------------------------------------------------
def _reader_worker(rank, world, port, scp, P, out_dir):
    """One rank of a data-parallel epoch: its share of the shuffled minibatches from ``io.RandExample(rank=, world=)``
    through ``parallel.chain_loss_data_parallel`` (both ranks on the test box's one GPU, the collective over gloo)."""
    import os

    import torch.distributed as dist

    from torchain_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fst = synth.random_den_fst(40, 4, P, seed=1)
        den = io.DenominatorGraph(fst, P)
        rd = io.RandExample(scp, seed=11, batchsize=2, prefetch=2, rank=rank, world=world)
        rows = []
        for step, ((inp, aux), sup) in enumerate(rd):
            B, T, _ = sup.shape
            torch.manual_seed(1000 + step)  # (the same "model output" on both ranks would hide a mix-up: seed by step AND rank)
            x = torch.randn(B, P, T, generator=torch.Generator().manual_seed(7 * step + rank)).cuda().requires_grad_(True)
            loss, res = parallel.chain_loss_data_parallel(x, den, sup, 1e-4, 0.1)
            loss.backward()
            rows.append([float(v) for v in res.data] + [float(loss), B * T, float(x.grad.abs().sum())])
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.asarray(rows, np.float64))
        with open(os.path.join(out_dir, "keys%d.txt" % rank), "w") as f:
            for i in range(rd.n_batch):
                f.write(" ".join(rd.batch_keys(i)) + "\n")
    finally:
        dist.destroy_process_group()


def test_two_ranks_fed_from_the_native_reader(tmp_path):
    """the rank-aware data path.  Two processes read their shares of one epoch
    (``tc_rand_reader_*`` with rank / world), run the data-parallel loss on them and must agree, step by step, on the
    GLOBAL results -- whose weight is the sum of the two ranks' frames -- while together covering every minibatch of
    the one-process list once."""
    import socket
    import sys

    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(__file__))
    from test_egs import _write_set

    P = 24
    fst = synth.random_den_fst(40, 4, P, seed=1)
    lengths = [5] * 8 + [8] * 6 + [11] * 2
    keyed, ark, scp = _write_set(tmp_path, fst, lengths)
    io.print_key_length("scp:" + scp, scp + ".len")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_reader_worker, args=(2, port, scp, P, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(str(tmp_path / ("rank%d.npy" % k))) for k in range(2)]
    assert r[0].shape == r[1].shape and r[0].shape[0] == 4  # 4 + 3 + 1 = 8 batches, four steps per rank
    np.testing.assert_array_equal(r[0][:, :4], r[1][:, :4])  # objf, l2_term, weight, loss: global, identical on both ranks
    np.testing.assert_array_equal(r[0][:, 2], r[0][:, 4] + r[1][:, 4])  # weight = frames of both ranks' batches (w = 1)
    assert (r[0][:, 5] > 0).all() and (r[1][:, 5] > 0).all()
    whole = io.RandExample(scp, seed=11, batchsize=2, prefetch=False)
    full = [" ".join(whole.batch_keys(i)) for i in range(whole.n_batch)]
    got = [open(str(tmp_path / ("keys%d.txt" % k))).read().split("\n")[:-1] for k in range(2)]
    assert got[0] == full[0::2] and got[1] == full[1::2]


@pytest.mark.parametrize("three_d", [True, False])
@pytest.mark.parametrize("kaldi_way", [True, False])
def test_one_call_step_numerical_failure_is_soft(three_d, kaldi_way):
    """[K] NaN objf -> objf = -10 * weight, the MMI derivative is the l2 term alone, xent_deriv zero -- through
    ``tc_chain_step``'s own ways of clearing the cross-entropy gradient (zero rows written by the denominator kernel, a
    cleared (B, C, T) tensor with entries written in place, the single call of the reference's two-call form) and with
    its cross-entropy objective, which must be that of a zero xent_deriv."""
    from torchain_amd.functions import chain_loss
    fst = synth.random_den_fst(300, 4, 96, seed=4)
    S, T, P = 5, 11, 96
    den = io.DenominatorGraph(fst, P)
    sup = synth.random_supervision(fst, S, T, 3, seed=2, initial_probs=den.initial_probs())
    hsup = io.Supervision.from_synth(sup)
    y = synth.random_nnet_output(S, T, P, seed=6)
    xe = synth.random_nnet_output(S, T, P, seed=7)
    (xe if not kaldi_way else y)[17, 5] = np.nan  # (the reference's way evaluates the objective on xent_input)
    a, b = torch.from_numpy(y).cuda(), torch.from_numpy(xe).cuda()
    if three_d:
        a = a.reshape(T, S, P).permute(1, 2, 0).contiguous()
        b = b.reshape(T, S, P).permute(1, 2, 0).contiguous()
    a.requires_grad_(True)
    b.requires_grad_(True)
    l2 = 1e-3
    loss, res = chain_loss(a, den, hsup, l2, 0.05, 0.1, b, kaldi_way)
    loss.backward()
    assert float(res.data[0]) == -10.0 * S * T and float(res.data[2]) == S * T
    src = b if not kaldi_way else a  # the tensor the objective was evaluated on
    finite = torch.isfinite(src.detach())
    # backward returns -deriv = +weight * l2 * output where the output is finite
    want = (sup.weight * l2) * src.detach()
    assert torch.allclose(a.grad[finite], want[finite], rtol=1e-6, atol=0)
    assert torch.count_nonzero(b.grad) == 0
    assert res.xent_objf == 0.0
