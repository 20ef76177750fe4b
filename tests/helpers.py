"""Shared helpers for the GPU parity tests: run the HIP path through the C ABI and compare with the
CPU oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import torch

from torchain_amd import io
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
