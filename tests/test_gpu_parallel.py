"""Data-parallel use on the GPU box's one device: ``parallel.chain_loss_data_parallel`` across two ranks (gloo between two
processes that both compute on device 0), two ranks fed by ``io.RandExample(rank, world)``, the RCCL branch of the one
all-reduce with a single rank, and BASELINE.json configs[3]'s 2048 sequences on one GPU through slice identities.
SURVEY.md section 8e; reference intent: ``example/chime5/parallel_train.py:70-75``."""
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


def _dp_worker(rank, world, port, S, T, P, out_dir):
    """One rank: its shard of the sequences through parallel.chain_loss_data_parallel on cuda:0 (both ranks share the
    GPU of the test box; the collective travels over gloo, the path every backend but RCCL takes)."""
    import torch.distributed as dist

    from torchain_amd import parallel
    from torchain_amd.synth import SupFst  # noqa: F401

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = np.load(os.path.join(out_dir, "inputs.npz"))
        fst = synth.random_den_fst(120, 5, P, seed=71)
        sup = synth.SupFst(float(d["w"]), S, T, P, int(d["nst"]), d["arc_begin"], d["ilabel"], d["arc_weight"], d["nextstate"], d["final"])
        lo, hi = parallel.shard_range(S, rank, world)
        y_local = np.ascontiguousarray(parallel.shard_rows(torch.from_numpy(d["y"]), S, lo, hi).numpy())
        xe_local = np.ascontiguousarray(parallel.shard_rows(torch.from_numpy(d["xe"]), S, lo, hi).numpy())
        sup_local = parallel.shard_supervision_fst(sup, lo, hi)
        den, hsup = io.DenominatorGraph(fst, P), io.Supervision.from_synth(sup_local)
        B = hi - lo
        x = to3d(y_local, B, T, P).requires_grad_(True)
        x2 = to3d(xe_local, B, T, P).requires_grad_(True)
        loss, res = parallel.chain_loss_data_parallel(x, den, hsup, 1e-3, 0.1, 0.1, x2, kaldi_way=True)
        loss.backward()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), res=res.data.numpy(), xent=res.xent_objf, loss=float(loss),
                 grad=from3d(x.grad, B, T, P), xgrad=from3d(x2.grad, B, T, P))
    finally:
        dist.destroy_process_group()


def test_chain_loss_data_parallel_two_ranks(oracle, tmp_path):
    """example/chime5/parallel_train.py:59-75 done right: every rank holds the GLOBAL [objf, l2_term, weight] and xent
    objective after ONE all-reduce, its loss is the global -objf/weight, and its gradient rows are the full batch's
    rows of its own sequences -- with the product function on both ranks."""
    import torch.multiprocessing as mp

    from torchain_amd import parallel

    world, S, T, P = 2, 5, 12, 64  # uneven shards: 3 + 2 sequences
    fst = synth.random_den_fst(120, 5, P, seed=71)
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 1, seed=72, weight=0.5, initial_probs=g.initial_probs())  # 1 path: single boundary states
    y = synth.random_nnet_output(S, T, P, seed=73)
    xe = torch.log_softmax(torch.from_numpy(synth.random_nnet_output(S, T, P, seed=74)), dim=1).numpy()
    np.savez(str(tmp_path / "inputs.npz"), y=y, xe=xe, w=sup.weight, nst=sup.num_states, arc_begin=sup.arc_begin,
             ilabel=sup.ilabel, arc_weight=sup.arc_weight, nextstate=sup.nextstate, final=sup.final)
    mp.spawn(_dp_worker, args=(world, free_port(), S, T, P, str(tmp_path)), nprocs=world, join=True)
    full = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-3, 0.1, want_xent=True)
    want_xent = float((xe.astype(np.float64) * full["xent_deriv"].astype(np.float64)).sum())
    r = [np.load(str(tmp_path / ("rank%d.npz" % k))) for k in range(world)]
    np.testing.assert_array_equal(r[0]["res"], r[1]["res"])  # every rank holds the same global results
    assert r[0]["loss"] == r[1]["loss"]
    assert abs(r[0]["res"][0] - full["objf"]) <= REL * abs(full["objf"])
    assert abs(r[0]["res"][1] - full["l2_term"]) <= REL * abs(full["l2_term"])
    assert r[0]["res"][2] == full["weight"]
    assert abs(float(r[0]["xent"]) - want_xent) <= REL * abs(want_xent) and float(r[0]["xent"]) == float(r[1]["xent"])
    assert abs(r[0]["loss"] - (-full["objf"] / full["weight"])) <= REL * abs(full["objf"] / full["weight"])
    for k in range(world):
        lo, hi = parallel.shard_range(S, k, world)
        want = parallel.shard_rows(torch.from_numpy(full["deriv"]), S, lo, hi).numpy()
        wantx = parallel.shard_rows(torch.from_numpy(full["xent_deriv"]), S, lo, hi).numpy()
        assert rel_err(r[k]["grad"], -want, floor=0.5) <= REL
        assert rel_err(r[k]["xgrad"], -0.1 * wantx, floor=0.05) <= REL


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
    from fixtures import write_set as _write_set

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


def test_xent_objective_value_and_rccl_branch(oracle):
    """(a) ChainResults.xent_objf = sum(xent_output * w * numerator posteriors), the cross-entropy objective
    Kaldi's chain trainer logs (a TODO in the reference, torchain/functions.py:88-89), against the oracle's
    xent derivative; (b) chain_loss_data_parallel through the REAL RCCL path with a process group of one rank
    (the recipe's per-device loss, example/chime5/parallel_train.py:59-75: results summed over devices,
    loss = -sum(objf) / sum(weight))."""
    import os
    import torch.distributed as dist
    from torchain_amd import io, parallel
    from torchain_amd.functions import chain_loss

    fst = synth.random_den_fst(120, 5, 64, seed=71)
    B, T, P = 4, 15, 64
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 3, seed=72, weight=0.5, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=73)
    xe = torch.log_softmax(torch.from_numpy(synth.random_nnet_output(B, T, P, seed=74)), dim=1).numpy()
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    want_xent_objf = float((xe.astype(np.float64) * ref["xent_deriv"].astype(np.float64)).sum())
    den, hsup = io.DenominatorGraph(fst, P), io.Supervision.from_synth(sup)
    to3d = lambda a: torch.from_numpy(a.reshape(T, B, P).transpose(1, 2, 0).copy()).cuda()
    loss, res = chain_loss(to3d(y).requires_grad_(True), den, hsup, 1e-4, 0.1, 0.1, to3d(xe).requires_grad_(True),
                           kaldi_way=True)
    assert abs(res.xent_objf - want_xent_objf) <= REL * abs(want_xent_objf)
    assert abs(res.xent_loss - (-want_xent_objf / ref["weight"])) <= REL * abs(want_xent_objf / ref["weight"])
    _, res0 = chain_loss(to3d(y), den, hsup, 1e-4, 0.1)
    assert res0.xent_objf is None and res0.xent_loss is None

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        loss2, res2 = parallel.chain_loss_data_parallel(to3d(y).requires_grad_(True), den, hsup, 1e-4, 0.1, 0.1,
                                                        to3d(xe).requires_grad_(True), kaldi_way=True, even_if_alone=True)
        assert dist.get_backend() == "nccl"
        assert torch.equal(res2.data, res.data) and float(loss2) == float(loss)
        assert abs(res2.xent_objf - res.xent_objf) <= 1e-9 * abs(res.xent_objf)
    finally:
        dist.destroy_process_group()


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
