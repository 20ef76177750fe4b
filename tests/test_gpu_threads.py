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
