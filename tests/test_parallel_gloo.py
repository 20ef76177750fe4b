"""The N > 1 path on CPU: two processes, gloo backend.  Each rank owns a contiguous shard of the
sequences, evaluates the chain objective on its shard (here with the CPU oracle standing in for the
per-GPU HIP call -- the oracle is the checker, the thing under test is the sharding and the
collective), then ONE all-reduce of (objf, l2_term, weight, xent_objf) must reproduce the full-batch result, and
each rank's derivative rows must equal the corresponding rows of the full-batch derivative."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torchain_amd import parallel, synth
from torchain_amd.functions import ChainResults


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, S, T, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle

        fst = synth.random_den_fst(40, 4, 24, seed=31)
        g = pyoracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 1, seed=32, initial_probs=g.initial_probs())  # 1 path: single boundary states
        y = synth.random_nnet_output(S, T, 24, seed=33)
        lo, hi = parallel.shard_range(S, rank, world)
        y_local = np.ascontiguousarray(parallel.shard_rows(torch.from_numpy(y), S, lo, hi).numpy())
        sup_local = parallel.shard_supervision_fst(sup, lo, hi)
        local = pyoracle.compute_chain_objf_and_deriv(g, sup_local, y_local, 1e-3, 0.1)
        res = ChainResults()
        res.data[:] = torch.from_numpy(local["results"])
        res.xent_objf = 1.5 + rank  # (a stand-in: the fourth value of the one collective)
        parallel.all_reduce_results(res)
        assert res.xent_objf == 1.5 + 2.5
        np.save(os.path.join(out_dir, "res%d.npy" % rank), res.data.numpy())
        np.save(os.path.join(out_dir, "deriv%d.npy" % rank), local["deriv"])
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce(tmp_path, oracle):
    world, S, T = 2, 5, 8  # uneven shards: 3 + 2 sequences
    port = _free_port()
    mp.spawn(_worker, args=(world, port, S, T, str(tmp_path)), nprocs=world, join=True)
    fst = synth.random_den_fst(40, 4, 24, seed=31)
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 1, seed=32, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, 24, seed=33)
    full = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-3, 0.1)
    r0 = np.load(str(tmp_path / "res0.npy"))
    r1 = np.load(str(tmp_path / "res1.npy"))
    np.testing.assert_array_equal(r0, r1)  # every rank holds the same global results
    assert abs(r0[0] - full["objf"]) <= 1e-5 * abs(full["objf"])
    assert abs(r0[1] - full["l2_term"]) <= 1e-5 * abs(full["l2_term"])
    assert r0[2] == full["weight"] == S * T
    for rank in range(world):
        lo, hi = parallel.shard_range(S, rank, world)
        want = parallel.shard_rows(torch.from_numpy(full["deriv"]), S, lo, hi).numpy()
        got = np.load(str(tmp_path / ("deriv%d.npy" % rank)))
        assert np.abs(got - want).max() <= 1e-5


def test_combine_results_matches_sum():
    a, b = ChainResults(), ChainResults()
    a.data[:] = torch.tensor([-10.0, -1.0, 20.0])
    b.data[:] = torch.tensor([-30.0, -2.0, 28.0])
    c = parallel.combine_results([a, b])
    assert torch.equal(c.data, torch.tensor([-40.0, -3.0, 48.0]))
    assert float(c.loss) == pytest.approx(40.0 / 48.0)
