"""Two host threads driving the same graph on streams of their own: the per-device state of the library (supervision
pool, graph tables, the side stream and fork / join events of the two-CU form) is shared, the results must not be."""
import threading

import numpy as np
import pytest
import torch

from torchain_amd import io, synth

from helpers import hip_den

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_two_threads_two_streams_one_graph(kernel_family, form):
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("C2")
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    S, T = 3, 25
    ys = [synth.random_nnet_output(S, T, fst.num_pdfs, seed=40 + i, scale=2.0) for i in range(2)]
    want = [hip_den(fst, y, S, leaky=0.1, deriv_weight=-1.0, l2_scale=1e-4, graph=graph) for y in ys]
    got = [[] for _ in ys]
    errors = []

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(20):
                    got[i].append(hip_den(fst, ys[i], S, leaky=0.1, deriv_weight=-1.0, l2_scale=1e-4, graph=graph))
        except Exception as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for out in got[i]:
            assert out["status"] == 0 and out["logprob"] == want[i]["logprob"]
            assert np.array_equal(out["deriv"], want[i]["deriv"])


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_back_to_back_calls_on_two_streams_without_host_sync(kernel_family, form):
    """Calls on alternating streams, enqueued without waiting for one another (each with its own buffers): the
    library's shared side stream and fork / join events must order every call's pieces among themselves only."""
    import ctypes as C
    from torchain_amd._lib import check, lib
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("C2")
    graph = io.DenominatorGraph(fst, fst.num_pdfs).prepare(torch.device("cuda", 0))
    S, T, P = 2, 40, fst.num_pdfs
    n = 12
    ys = [torch.from_numpy(synth.random_nnet_output(S, T, P, seed=60 + i, scale=2.0)).cuda() for i in range(n)]
    nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)

    def enqueue(y, stream):
        with torch.cuda.stream(stream):  # (the zero fill of lp must be ordered before the call that writes it)
            d = torch.empty_like(y)
            lp = torch.zeros(1, dtype=torch.float64, device="cuda")
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        check(lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), 0.1, -1.0, 1e-4, 0,
            C.c_void_p(d.data_ptr()), d.stride(0), C.c_void_p(lp.data_ptr()), None, C.c_void_p(ws.data_ptr()), nbytes, 0,
            C.c_void_p(stream.cuda_stream)), "den")
        return d, lp, ws

    main = torch.cuda.current_stream()
    want = []
    for y in ys:  # one at a time
        d, lp, _ = enqueue(y, main)
        torch.cuda.synchronize()
        want.append((d.clone(), float(lp)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got = [enqueue(y, streams[i % 2]) for i, y in enumerate(ys)]  # all in flight together
    torch.cuda.synchronize()
    for (d, lp, _), (wd, wlp) in zip(got, want):
        assert float(lp) == wlp and torch.equal(d, wd)
