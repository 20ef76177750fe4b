"""BASELINE.json's configurations at FULL size against the CPU oracle (round-1 gap: they were only compared at
toy batch sizes).  The oracle runs the whole C3 denominator in ~5 s and C2's full objective in ~2 s, so long-T
drift (150 renormalised frames), all 256 workgroups and the API-default leaky coefficient at T = 150 are all
inside the comparison.  Tolerance: 1e-4 relative (north_star), as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from torchain_amd import io, synth

from helpers import hip_chain, hip_den, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


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


@pytest.mark.parametrize("leaky", [0.1, 1e-5])
def test_config3_full_size_denominator(oracle, leaky):
    """configs[2], the metric's workload: batch 256 x 150 frames x 4096 pdfs; log-prob and the whole derivative.
    leaky = 1e-5 is the API default (torchain/functions.py:128-130) at the full 150 frames."""
    c = synth.CONFIGS["C3"]
    fst = synth.config_den_fst("C3")
    S, T, P = c["S"], c["T"], c["P"]
    y = synth.random_nnet_output(S, T, P, seed=1237)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
    assert out["status"] == 0 and ref["ok"]
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"]) <= REL
    # element-wise on everything that is not rounding dust: the per-state (subtraction) form of gamma on tied graphs
    big = ref["deriv"] > 1e-4
    assert (np.abs(out["deriv"][big] - ref["deriv"][big]) / ref["deriv"][big]).max() <= 1e-3


def test_config5_full_size_denominator(oracle):
    """configs[4]: large-vocabulary graph (P=10240, A=61440), batch 128 x 150, leaky 0.1: the tight LDS layout."""
    c = synth.CONFIGS["C5"]
    fst = synth.config_den_fst("C5")
    S, T, P = c["S"], c["T"], c["P"]
    y = synth.random_nnet_output(S, T, P, seed=1239)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=c["leaky"], deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 1
    assert out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"], ref["deriv"]) <= REL


def test_config4_per_node_size_2048_sequences_on_one_gpu():
    """configs[3]'s per-node problem (2048 sequences) on ONE GPU: sequences never interact and results are
    bitwise reproducible, so every 256-sequence slice of the big batch must reproduce, bit for bit, the rows a
    separate 256-sequence call gives, and the batch log-prob is the sum of the slices'.  Everything stays on the
    device (5 GB of nnet output)."""
    import ctypes as C
    from torchain_amd._lib import check, lib

    c = synth.CONFIGS["C4"]
    fst = synth.config_den_fst("C4")
    S, T, P = c["S"], c["T"], c["P"]
    dev = torch.device("cuda", 0)
    graph = io.DenominatorGraph(fst, P).prepare(dev)
    gen = torch.Generator(device=dev).manual_seed(2048)
    y = torch.randn(T, S, P, device=dev, generator=gen)
    stream = torch.cuda.current_stream().cuda_stream

    def den(y2d, nseq):
        deriv = torch.empty_like(y2d)
        nbytes = lib.tc_chain_workspace_bytes(graph.ptr, nseq, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        lp = torch.zeros(1, dtype=torch.float64, device=dev)
        st = torch.full((1,), -1, dtype=torch.int32, device=dev)
        rc = lib.tc_den_forward_backward(
            graph.ptr, nseq, C.c_void_p(y2d.data_ptr()), nseq * T, P, y2d.stride(0), c["leaky"], -1.0, c["l2"], 0,
            C.c_void_p(deriv.data_ptr()), deriv.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(st.data_ptr()),
            C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream))
        check(rc, "tc_den_forward_backward")
        torch.cuda.synchronize()
        return deriv, float(lp.item()), int(st.item())

    full, lp_full, st_full = den(y.view(T * S, P), S)
    assert st_full == 0
    rows = (full.view(T, S, P) + c["l2"] * y).sum(dim=2)  # = -sum_pdf gamma = -1 per (frame, sequence)
    assert float((rows + 1.0).abs().max()) < 1e-3
    lp_sum = 0.0
    for lo in (0, 768, 1792):
        part = y[:, lo:lo + 256, :].contiguous()
        d, lp, st = den(part.view(T * 256, P), 256)
        assert st == 0
        assert torch.equal(d.view(T, 256, P), full.view(T, S, P)[:, lo:lo + 256, :])
        lp_sum += lp
    # log-probs of the three slices against the same slices of the big batch (per-sequence values are not
    # exposed; the slices are summed by the same fixed-order reduction)
    d3, lp3, _ = den(torch.cat([y[:, lo:lo + 256, :] for lo in (0, 768, 1792)], dim=1).contiguous().view(T * 768, P), 768)
    assert abs(lp3 - lp_sum) <= 1e-9 * abs(lp_sum)
    assert abs(lp_full) > abs(lp3)
