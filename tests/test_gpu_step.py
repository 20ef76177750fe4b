"""The training-side step around the hot call (``chain_loss`` -> ``tc_chain_step``) and its evaluation form.

* EVALUATION STEP: ``chain_loss`` under ``torch.no_grad()`` (the recipe's validation loop,
  ``/root/reference/example/chime5/train.py:150-171``) takes ``tc_chain_step(grad = NULL)``: [K] ``ComputeChainObjfAndDeriv``
  with ``nnet_output_deriv == NULL`` -- forward recursions only.  The reference has no ``needs_input_grad`` check
  (``torchain/functions.py:74,82``) and pays for a training step there; the values it reports are the same, which is what these
  tests hold the evaluation step to (``results`` equal to the training call's to 1e-6, in both layouts, on C2 and on a graph of
  the R4 class), together with the library's launch counters showing that no backward recursion was enqueued.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from torchain_amd import io, synth
from torchain_amd._lib import check, lib
from torchain_amd.functions import ChainResults, chain_loss

pytestmark = pytest.mark.gpu
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
