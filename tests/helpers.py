"""Shared helpers for the GPU parity tests: run the HIP path through the C ABI and compare with the
CPU oracle on the same seeded inputs."""
import ctypes as C
import functools
import os
import socket

import numpy as np
import torch

from torchain_amd import io, synth
from torchain_amd._lib import check, lib
from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(max |b|, floor)  (the '1e-4 relative' of BASELINE.json's north_star,
    matrix-wise).  ``floor`` is the natural scale of the quantity when the reference itself is a
    difference of larger terms: the derivative is w*(gamma_num - gamma_den) with both posteriors in
    [0, 1], so its scale is the supervision weight even where the two cancel."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))


def hip_chain(fst, sup, y, l2=0.0, leaky=1e-5, xent=False, want_deriv=True, device="cuda:0", graph=None,
              row_pad=0):
    """Runs tc_chain_objf_and_deriv; returns dict(results, deriv, xent_deriv) as numpy."""
    graph = graph or io.DenominatorGraph(fst, fst.num_pdfs)
    hsup = io.Supervision.from_synth(sup)
    rows, cols = y.shape
    yt_full = torch.zeros(rows, cols + row_pad, device=device)
    yt_full[:, :cols] = torch.from_numpy(np.ascontiguousarray(y)).to(device)
    yt = yt_full[:, :cols]
    deriv = torch.full((rows, cols + row_pad), 7.0, device=device)[:, :cols] if want_deriv else None
    xd = torch.full((rows, cols + row_pad), 7.0, device=device)[:, :cols] if xent else None
    res = ChainResults()
    compute_chain_objf_and_deriv(graph, hsup, yt, res.data, deriv, xd, l2, leaky, 0.1 if xent else 0.0)
    torch.cuda.synchronize()
    return dict(results=res.data.numpy().copy(), deriv=None if deriv is None else deriv.cpu().numpy(),
                xent_deriv=None if xd is None else xd.cpu().numpy(), graph=graph)


def hip_den(fst, y, S, leaky=1e-5, deriv_weight=1.0, l2_scale=0.0, accumulate=False, want_deriv=True,
            device="cuda:0", graph=None, init=None):
    """Runs tc_den_forward_backward; returns dict(logprob, deriv, status)."""
    graph = graph or io.DenominatorGraph(fst, fst.num_pdfs)
    rows, cols = y.shape
    yt = torch.from_numpy(np.ascontiguousarray(y)).to(device)
    if want_deriv:
        deriv = torch.full((rows, cols), 3.0 if init is None else float(init), device=device)
    else:
        deriv = None
    T = rows // S
    nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    lp = torch.zeros(1, dtype=torch.float64, device=device)
    st = torch.full((1,), -1, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(yt.data_ptr()), rows, cols, yt.stride(0), leaky, deriv_weight, l2_scale,
        1 if accumulate else 0, C.c_void_p(deriv.data_ptr()) if want_deriv else None, cols,
        C.c_void_p(lp.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes,
        torch.cuda.current_device(), C.c_void_p(stream))
    check(rc, "tc_den_forward_backward")
    torch.cuda.synchronize()
    return dict(logprob=float(lp.item()), deriv=None if deriv is None else deriv.cpu().numpy(), status=int(st.item()),
                graph=graph)


def hip_num(sup, y, want_deriv=True, device="cuda:0"):
    """Runs tc_num_forward_backward; returns dict(logprob_weighted, deriv)."""
    hsup = io.Supervision.from_synth(sup)
    rows, cols = y.shape
    yt = torch.from_numpy(np.ascontiguousarray(y)).to(device)
    deriv = torch.zeros(rows, cols, device=device) if want_deriv else None
    nbytes = 256 * ((hsup.n_batch * 8 + 255) // 256) + 256
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    lp = torch.zeros(1, dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.tc_num_forward_backward(
        hsup.ptr, C.c_void_p(yt.data_ptr()), rows, cols, yt.stride(0),
        C.c_void_p(deriv.data_ptr()) if want_deriv else None, cols, C.c_void_p(lp.data_ptr()),
        C.c_void_p(ws.data_ptr()), nbytes, torch.cuda.current_device(), C.c_void_p(stream))
    check(rc, "tc_num_forward_backward")
    torch.cuda.synchronize()
    return dict(logprob_weighted=float(lp.item()), deriv=None if deriv is None else deriv.cpu().numpy())


REL = 1e-4  # north_star: "within 1e-4 relative"


# ---- checkers shared by the GPU test modules (moved here from the round-numbered modules in round 6) ---------------------------
def check_full(oracle, fst, S, T, l2, leaky, weight=1.0, seed=5, zero=False, row_pad=0, paths=3):
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


def peaky_elem(got, ref, lo):
    m = ref > lo
    return float((np.abs(got[m] - ref[m]) / ref[m]).max()) if m.any() else 0.0


def peaky_check(oracle, fst, S, T, scale, leaky, beyond_clamp=False):
    from oracle import independent_f64 as ind

    g = oracle.DenGraph(fst)
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=11, scale=scale)
    if beyond_clamp:
        y[::7] *= 4.0
        assert np.abs(y).max() > 30.0
    lp, gam = ind.den_logprob_and_deriv(fst, g.initial_probs(), np.clip(y, -30.0, 30.0), S, leaky)
    ref = oracle.den_forward_backward(g, y, S, leaky=leaky, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
    assert out["status"] == 0
    assert abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
    d = out["deriv"]
    assert np.abs(d - gam).max() <= 1e-5                       # absolute: posteriors live in [0, 1]
    assert np.abs(d.sum(axis=1, dtype=np.float64) - 1.0).max() <= 1e-5
    # element-wise relative error by magnitude class: what the subtraction and the 2^-31 fixed point cost
    assert peaky_elem(d, gam, 1e-2) <= 1e-4
    assert peaky_elem(d, gam, 1e-4) <= 3e-3
    assert peaky_elem(d, gam, 1e-6) <= 0.15
    # the distance to the Kaldi-style oracle is the oracle's own distance to the truth (plus rounding)
    assert np.abs(d - ref["deriv"]).max() <= np.abs(ref["deriv"] - gam).max() + 1e-5
    assert abs(out["logprob"] - ref["logprob"]) <= 1e-4 * abs(ref["logprob"])


def to3d(a, B, T, P):
    return torch.from_numpy(a.reshape(T, B, P).transpose(1, 2, 0).copy()).cuda()


def from3d(g, B, T, P):
    return g.permute(2, 0, 1).reshape(T * B, P).cpu().numpy()


# ---- chain_loss_data_parallel across two ranks -----------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def elementwise(got, ref, what, bounds=((1e-3, 1e-4), (1e-4, 1e-3))):
    """entries of |ref| > 1e-3 within 1e-4 relative, entries > 1e-4 within 1e-3 relative"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    for floor, tol in bounds:
        m = np.abs(ref) > floor
        assert m.any(), (what, floor)
        worst = float((np.abs(got[m] - ref[m]) / np.abs(ref[m])).max())
        assert worst <= tol, (what, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))


def occupy_half_the_cus(stream, millis):
    """A long kernel on ``stream`` that holds half of the CUs: torch kernels sized to the device, far more work than
    the launches under test (a matmul chain of ~`millis` ms on 128 of the 256 CUs' worth of workgroups)."""
    n = 2048
    a = torch.randn(n, n, device="cuda")
    with torch.cuda.stream(stream):
        x = a
        for _ in range(max(1, millis // 2)):
            x = torch.tanh(x @ a * 1e-3)
    return x


def oracle_den(oracle, fst, y, S, T, leaky):
    threads = max(1, min(64, os.cpu_count() or 1))
    lp, deriv = oracle.den_forward_backward_blocks(oracle.DenGraph(fst), y, S, T, leaky, block=max(1, S // threads),
                                                   threads=threads, deriv_weight=1.0)
    return lp, deriv


def compare_at_size(oracle, cfg, S, T, seed, expect_tied):
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    y = synth.random_nnet_output(S, T, c["P"], seed=seed)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == expect_tied, out["graph"].stats()
    assert out["status"] == 0
    ref_lp, ref = oracle_den(oracle, fst, y, S, T, c["leaky"])
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp), (out["logprob"], ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "%s %dx%d" % (cfg, S, T))
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() <= 1e-4, np.abs(rows - 1.0).max()


@functools.lru_cache(maxsize=None)
def float64_truth(cfg, S, T, leaky):
    """(graph, outputs, float64 log-prob and occupation matrix) of a workload: shared by the kernel forms that are compared with it"""
    from oracle import independent_f64 as ind
    from oracle import pyoracle
    fst = synth.config_den_fst(cfg)
    pi = pyoracle.DenGraph(fst).initial_probs()
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=11)
    lp, gam = ind.den_logprob_and_deriv(fst, pi, np.clip(y, -30, 30), S, leaky)
    return fst, y, lp, np.asarray(gam, np.float64)
