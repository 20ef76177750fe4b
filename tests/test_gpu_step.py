"""The training-side step around the hot call -- ``chain_loss`` / ``_ChainLoss`` of ``torchain/functions.py:62-138`` -- as ONE library call
(``tc_chain_step``), and its evaluation form.

* wrapper semantics: ``(B, C, T)`` and 2-D inputs, ``backward`` = ``-deriv`` ignoring ``grad_output``, ``kaldi_way`` both ways, the fused
  layout kernels, the one-call step bit-identical to the multi-call wrappers, soft failure;
* EVALUATION STEP: ``chain_loss`` under ``torch.no_grad()`` (the recipe's validation loop, ``example/chime5/train.py:150-171``) takes
  ``tc_chain_step(grad = NULL)``: [K] ``ComputeChainObjfAndDeriv`` with ``nnet_output_deriv == NULL`` -- forward recursions only.  The
  reference has no ``needs_input_grad`` check (``torchain/functions.py:74,82``) and pays for a training step there; the values it
  reports are the same, which is what these tests hold the evaluation step to (``results`` equal to the training call's to 1e-6,
  both layouts, C2 and an R4-class graph), with the library's launch counters showing that no backward recursion was enqueued."""
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
from torchain_amd.functions import ChainResults, chain_loss

DEV = "cuda:0"


def _counters():
    return {k: int(lib.tc_debug_counter(k.encode())) for k in
            ("den_launches", "den_backward_launches", "num_launches", "num_backward_launches", "layout_launches")}


def _delta(before):
    now = _counters()
    return {k: now[k] - before[k] for k in now}


def _workload(cfgname, S, T, seed=3):
    cfg = synth.CONFIGS[cfgname]
    fst = synth.config_den_fst(cfgname)
    P = cfg["P"]
    graph = io.DenominatorGraph(fst, P).prepare(DEV)
    pi = graph.initial_probs()
    sup = synth.random_supervision(fst, S, T, 3, seed=seed, initial_probs=pi)
    y = synth.random_nnet_output(S, T, P, seed=seed + 1)
    return cfg, graph, io.Supervision.from_synth(sup), torch.from_numpy(y).to(DEV)


def _as_bct(y2d, S, T):
    P = y2d.shape[1]
    return y2d.view(T, S, P).permute(1, 2, 0).contiguous()


@pytest.mark.parametrize("cfgname,S,T", [("C2", 64, 150), ("R4", 32, 40)])
@pytest.mark.parametrize("three_d", [False, True])
def test_evaluation_step_equals_training_step(cfgname, S, T, three_d):
    cfg, graph, sup, y2d = _workload(cfgname, S, T)
    x = _as_bct(y2d, S, T) if three_d else y2d
    kw = dict(l2_regularize=cfg.get("l2", 0.0), leaky_hmm_coefficient=cfg["leaky"])
    xt = x.clone().requires_grad_(True)
    before = _counters()
    loss_t, res_t = chain_loss(xt, graph, sup, **kw)
    loss_t.backward()
    d = _delta(before)
    assert d["den_launches"] == 1 and d["den_backward_launches"] == 1 and d["num_backward_launches"] >= 1, d
    train = res_t.data.numpy().copy()

    before = _counters()
    with torch.no_grad():
        loss_e, res_e = chain_loss(x, graph, sup, **kw)
    d = _delta(before)
    assert d["den_launches"] == 1 and d["den_backward_launches"] == 0, d
    assert d["num_launches"] == 1 and d["num_backward_launches"] == 0, d
    assert d["layout_launches"] == (1 if three_d else 0), d  # (B, C, T): the frame-major copy in, nothing back
    ev = res_e.data.numpy()
    assert ev[2] == train[2]
    np.testing.assert_allclose(ev[:2], train[:2], rtol=1e-6)
    assert abs(float(loss_e) - float(loss_t)) <= 1e-6 * abs(float(loss_t))
    assert not loss_e.requires_grad and loss_e.device.type == "cuda"
    # an input that does not require grad, outside no_grad: the same route
    before = _counters()
    _, res_p = chain_loss(x, graph, sup, **kw)
    assert _delta(before)["den_backward_launches"] == 0
    np.testing.assert_array_equal(res_p.data.numpy(), ev)


@pytest.mark.parametrize("kaldi_way", [True, False])
@pytest.mark.parametrize("three_d", [False, True])
def test_evaluation_step_with_xent_branch(kaldi_way, three_d):
    """With the regulariser's branch the evaluation step still reports the cross-entropy objective (Kaldi's diagnostics do):
    the numerator's posteriors are formed, nothing dense is written.  ``kaldi_way=False`` reports the second call's
    results (on ``xent_input``), as the reference does (``torchain/functions.py:96-103``)."""
    S, T = 16, 30
    cfg, graph, sup, y2d = _workload("C2", S, T, seed=11)
    xe2d = torch.from_numpy(synth.random_nnet_output(S, T, cfg["P"], seed=99)).to(DEV)
    x, xe = (_as_bct(y2d, S, T), _as_bct(xe2d, S, T)) if three_d else (y2d, xe2d)
    kw = dict(l2_regularize=5e-5, leaky_hmm_coefficient=0.1, xent_regularize=0.1, kaldi_way=kaldi_way)
    xt, xet = x.clone().requires_grad_(True), xe.clone().requires_grad_(True)
    loss_t, res_t = chain_loss(xt, graph, sup, xent_input=xet, **kw)
    before = _counters()
    with torch.no_grad():
        loss_e, res_e = chain_loss(x, graph, sup, xent_input=xe, **kw)
    d = _delta(before)
    assert d["den_backward_launches"] == 0, d
    np.testing.assert_allclose(res_e.data.numpy(), res_t.data.numpy(), rtol=1e-6)
    assert abs(res_e.xent_objf - res_t.xent_objf) <= 1e-6 * abs(res_t.xent_objf)
    assert abs(float(loss_e) - float(loss_t)) <= 1e-6 * abs(float(loss_t))


def test_evaluation_step_soft_failure_and_c_abi():
    """grad == NULL through the bare C ABI; a NaN in the input still fails softly (objf = -10 * weight); a non-NULL xent_grad
    with a NULL grad is refused."""
    S, T = 8, 20
    cfg, graph, sup, y2d = _workload("C2", S, T, seed=5)
    P = cfg["P"]
    stream = torch.cuda.current_stream().cuda_stream
    nbytes = lib.tc_chain_step_workspace_bytes(graph.ptr, S, T, 0, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    out = torch.zeros(6, device=DEV)

    def step(y, grad=None, xgrad=None):
        return lib.tc_chain_step(graph.ptr, sup.ptr, C.c_void_p(y.data_ptr()), None, 0, y.stride(0), 5e-5, 0.1, 0.0, 1,
                                 None if grad is None else C.c_void_p(grad.data_ptr()),
                                 None if xgrad is None else C.c_void_p(xgrad.data_ptr()), C.c_void_p(out.data_ptr()),
                                 C.c_void_p(out.data_ptr() + 12), None, C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream))

    check(step(y2d), "tc_chain_step(eval)")
    ev = out.cpu().numpy().copy()
    g = torch.empty_like(y2d)
    check(step(y2d, g), "tc_chain_step(train)")
    tr = out.cpu().numpy()
    np.testing.assert_allclose(ev[:4], tr[:4], rtol=1e-6)
    assert ev[2] == S * T and abs(ev[3] + ev[0] / ev[2]) <= 1e-6 * abs(ev[3])
    assert step(y2d, None, g) < 0
    bad = y2d.clone()
    bad[3, 7] = float("nan")
    check(step(bad), "tc_chain_step(eval, NaN)")
    assert out.cpu().numpy()[0] == -10.0 * S * T


def test_evaluation_results_accumulate_like_the_recipe():
    """``valid_result.data += results.data`` over steps (example/chime5/train.py:166-170) on evaluation steps."""
    S, T = 8, 20
    cfg, graph, sup, y2d = _workload("C2", S, T, seed=7)
    total = ChainResults()
    with torch.no_grad():
        for _ in range(3):
            _, r = chain_loss(y2d, graph, sup, leaky_hmm_coefficient=0.1)
            total.data += r.data
    assert total.data[2] == 3 * S * T and abs(float(total.loss) - float(r.loss)) <= 1e-6 * abs(float(r.loss))


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


@pytest.mark.parametrize("three_d", [True, False])
def test_kaldi_way_false_is_the_objective_on_xent_input(oracle, three_d):
    """torchain/functions.py:96-103: with ``kaldi_way=False`` the results and the gradient of ``input`` are those of a
    second call whose nnet output is ``xent_input``; the gradient of ``xent_input`` is ``-xent_regularize`` times that
    second call's xent derivative.  All of it against the oracle evaluated on ``xent_input``."""
    from torchain_amd.functions import chain_loss

    fst = synth.random_den_fst(200, 5, 90, seed=13)
    B, T, P = 4, 23, 90
    l2, leaky, xr = 1e-4, 0.05, 0.1
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 3, seed=4, weight=0.5, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=5)
    xe = synth.random_nnet_output(B, T, P, seed=6)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, xe, l2, leaky, want_xent=True)  # the SECOND call's input
    den, hsup = io.DenominatorGraph(fst, P), io.Supervision.from_synth(sup)
    if three_d:
        x, x2 = to3d(y, B, T, P).requires_grad_(True), to3d(xe, B, T, P).requires_grad_(True)
    else:
        x, x2 = torch.from_numpy(y).cuda().requires_grad_(True), torch.from_numpy(xe).cuda().requires_grad_(True)
    loss, res = chain_loss(x, den, hsup, l2, leaky, xr, x2, kaldi_way=False)
    loss.backward()
    got = res.data.numpy()
    assert abs(got[0] - ref["objf"]) <= REL * abs(ref["objf"])
    assert abs(got[1] - ref["l2_term"]) <= REL * abs(ref["l2_term"])
    assert got[2] == ref["weight"]
    assert abs(float(loss) - (-ref["objf"] / ref["weight"])) <= REL * abs(ref["objf"] / ref["weight"])
    gx = from3d(x.grad, B, T, P) if three_d else x.grad.cpu().numpy()
    gxe = from3d(x2.grad, B, T, P) if three_d else x2.grad.cpu().numpy()
    assert rel_err(gx, -ref["deriv"], floor=0.5) <= REL          # MMI gradient: replaced by the second call's
    assert rel_err(gxe, -xr * ref["xent_deriv"], floor=0.05) <= REL


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


@pytest.mark.parametrize("three_d", [False, True])
@pytest.mark.parametrize("xent", [False, True])
def test_registered_operator_equals_chain_loss(three_d, xent):
    """``torch.ops.torchain_amd.chain_step`` (torchain_amd/ops.py: the step as a ``torch.library`` operator with an autograd formula)
    against ``chain_loss``: loss, results, both gradients bit for bit; the evaluation step under ``no_grad``; ``grad_output`` ignored
    as in the reference's backward (``torchain/functions.py:106-115``)."""
    from torchain_amd.ops import chain_loss_op
    S, T = 8, 20
    cfg, graph, sup, y2d = _workload("C2", S, T, seed=21)
    xe2d = torch.from_numpy(synth.random_nnet_output(S, T, cfg["P"], seed=22)).to(DEV)
    x, xe = (_as_bct(y2d, S, T), _as_bct(xe2d, S, T)) if three_d else (y2d, xe2d)
    kw = dict(l2_regularize=5e-5, leaky_hmm_coefficient=0.1, xent_regularize=0.1 if xent else 0.0, kaldi_way=True)
    outs = []
    for fn in (chain_loss, chain_loss_op):
        a = x.clone().requires_grad_(True)
        b = xe.clone().requires_grad_(True) if xent else None
        loss, res = fn(a, graph, sup, xent_input=b, **kw)
        (3.0 * loss).sum().backward()  # (the factor changes nothing)
        outs.append((loss.detach().clone(), res.data.clone(), a.grad.clone(), None if b is None else b.grad.clone(), res.xent_objf))
    (l0, r0, g0, x0, o0), (l1, r1, g1, x1, o1) = outs
    assert torch.equal(l0, l1) and torch.equal(r0, r1) and torch.equal(g0, g1) and o0 == o1
    assert (x0 is None and x1 is None) or torch.equal(x0, x1)
    before = _counters()
    with torch.no_grad():
        le, re_ = chain_loss_op(x, graph, sup, xent_input=xe if xent else None, **kw)
    assert _delta(before)["den_backward_launches"] == 0
    np.testing.assert_allclose(re_.data.numpy(), r0.numpy(), rtol=1e-6)


def test_registered_operator_traces_as_one_node():
    """What the registration is for: a function whose loss is the operator compiles with ``fullgraph=True`` (fake implementation for
    the shapes, the registered autograd formula for the backward) and returns the eager path's loss and gradient bit for bit."""
    from torchain_amd import ops  # noqa: F401
    S, T = 4, 20
    fst = synth.random_den_fst(256, 6, 100, seed=5)
    graph = io.DenominatorGraph(fst, 100).prepare(DEV)
    sup = io.Supervision.from_synth(synth.random_supervision(fst, S, T, 3, seed=9, initial_probs=graph.initial_probs()))
    x = torch.from_numpy(synth.random_nnet_output(S, T, 100, seed=10)).to(DEV)
    gp, sp = int(graph.ptr.value), int(sup.ptr.value)

    def step(a):
        out, _grad, _xgrad = torch.ops.torchain_amd.chain_step(a, a.new_empty((0,)), gp, sp, 5e-5, 0.1, 0.0, True, True)
        return out[3]

    a = x.clone().requires_grad_(True)
    loss = torch.compile(step, backend="aot_eager", fullgraph=True)(a)
    loss.backward()
    b = x.clone().requires_grad_(True)
    ref, _ = chain_loss(b, graph, sup, 5e-5, 0.1)
    ref.backward()
    assert float(loss.detach()) == float(ref.detach()) and torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("three_d", [False, True])
def test_training_step_replays_from_a_captured_graph(three_d):
    """The captured step: ``chain_loss`` with the xent branch + the backward, captured once into a
    ``torch.cuda.CUDAGraph`` -- the library's side streams fork from and join the capturing stream, its pools are warm, nothing in the
    call reads the device -- and replayed on new activations: loss and both gradients bit-identical to the direct call's."""
    S, T = 8, 30
    cfg, graph, sup, y2d = _workload("C2", S, T)
    kw = dict(l2_regularize=5e-5, leaky_hmm_coefficient=cfg["leaky"], xent_regularize=0.1, kaldi_way=True)
    x = (_as_bct(y2d, S, T) if three_d else y2d).clone().requires_grad_(True)
    xe = (0.5 * x.detach()).clone().requires_grad_(True)

    seen = []

    def step():
        loss, results = chain_loss(x, graph, sup, xent_input=xe, **kw)
        g, gx = torch.autograd.grad(loss, (x, xe))
        seen.append(results)
        return loss.detach(), g, gx

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()  # warm-up outside the capture: tables, pools, side streams
    side.synchronize()
    before = _counters()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=side):
        captured = step()
    assert _delta(before)["den_launches"] == 1  # (captured, not run)
    captured_results = seen[-1]
    # a supervision the library has not seen before the capture: staged and uploaded by a node of the graph
    if not three_d:
        sup2 = io.Supervision.from_synth(synth.random_supervision(synth.config_den_fst("C2"), S, T, 3, seed=77,
                                                                  initial_probs=graph.initial_probs()))
        cg2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg2, stream=side):
            loss2, _ = chain_loss(x, graph, sup2, xent_input=xe, **kw)
            g2 = torch.autograd.grad(loss2, x)[0]
        cg2.replay()
        torch.cuda.synchronize()
        got2 = g2.clone()
        want_loss2, _ = chain_loss(x, graph, sup2, xent_input=xe, **kw)
        assert torch.equal(got2, torch.autograd.grad(want_loss2, x)[0])
    for seed in (21, 22):
        fresh = torch.randn(x.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        with torch.no_grad():
            x.copy_(fresh)
            xe.copy_(0.25 * fresh)
        before = _counters()
        cg.replay()
        torch.cuda.synchronize()
        assert _delta(before)["den_launches"] == 0  # no library call: the graph's own nodes ran
        got = [t.clone() for t in captured]
        captured_results.invalidate()  # the replay wrote new values behind it
        got_results = captured_results.data.clone(), captured_results.xent_objf
        want = step()
        torch.cuda.synchronize()
        assert torch.equal(got_results[0], seen[-1].data) and got_results[1] == seen[-1].xent_objf
        assert abs(float(got[0]) + float(got_results[0][0]) / float(got_results[0][2])) <= 1e-6 * abs(float(got[0]))
        assert bool(torch.isfinite(got[0]).all()) and float(got[1].abs().max()) > 0
        for a, b in zip(got, want):
            assert torch.equal(a, b)
