"""Data-parallel use of the chain loss: one process per GPU, shard the minibatch's sequences.

Sequences never interact in the chain objective: the denominator alpha/beta
are per sequence, the merged numerator FST factors per sequence, ``objf``, ``l2_term`` and ``weight``
are plain sums and every derivative row belongs to one sequence.  So rank r runs the unchanged
single-GPU path on its own sequences (its model replica produced their nnet output, so no
activation moves) and the only exchange is ONE all-reduce (SUM) of the three floats
``(objf, l2_term, weight)`` -- RCCL over xGMI when the process group's backend is ``"nccl"``; ``gloo``
works for CPU tests.  The derivative is never communicated (DDP reduces parameter gradients).

The reference has no working antecedent: ``example/chime5/train.py:107`` computes the loss on GPU 0
over the gathered batch and ``parallel_train.py:26-75`` (per-device loss) is broken; its intent --
``results.data = sum over devices``, ``loss = sum(loss_d * weight_d) / sum(weight_d)``
(``parallel_train.py:70-75``) -- is what ``all_reduce_results`` implements.
"""
import numpy as np
import torch

from .functions import ChainResults, _chain_loss_into


def shard_range(num_sequences, rank, world_size):
    """Contiguous block of sequences owned by ``rank``: [lo, hi).  Blocks differ by at most one."""
    base, rem = divmod(int(num_sequences), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(x2d, num_sequences, lo, hi):
    """Rows of a frame-major ``(T*S, C)`` matrix (row = t*S + s) that belong to sequences [lo, hi),
    as a ``(T*(hi-lo), C)`` matrix in the same frame-major order."""
    rows, cols = x2d.shape
    T = rows // num_sequences
    return x2d.reshape(T, num_sequences, cols)[:, lo:hi, :].reshape(T * (hi - lo), cols)


def shard_supervision_fst(sup, lo, hi):
    """Per-rank supervision from a ``torchain_amd.synth.SupFst``-like merged acceptor: keeps the
    sub-acceptor of sequences [lo, hi).  Host-side numpy; mirrors what a per-rank egs reader would
    deliver.  Requires one boundary state between consecutive sequences (true for supervisions whose
    sequences end in a single final state); otherwise build per-rank supervisions at the reader."""
    S, T = sup.num_sequences, sup.frames_per_sequence
    nst = sup.num_states
    times = np.full(nst, -1, np.int64)
    times[0] = 0
    for s in range(nst):
        for a in range(sup.arc_begin[s], sup.arc_begin[s + 1]):
            times[sup.nextstate[a]] = times[s] + 1
    first = np.searchsorted(times, [lo * T, lo * T + 1, hi * T, hi * T + 1])
    b0, b1, e0, e1 = (int(v) for v in first)
    if (b1 - b0 != 1) or (hi < S and e1 - e0 != 1):
        raise ValueError("sequence boundary is not a single state; shard at the egs reader instead")
    keep = np.arange(b0, e1)
    remap = {int(g): i for i, g in enumerate(keep)}
    arc_begin = [0]
    il, aw, nx = [], [], []
    for g in keep:
        if g < e0:
            for a in range(sup.arc_begin[g], sup.arc_begin[g + 1]):
                il.append(int(sup.ilabel[a]))
                aw.append(float(sup.arc_weight[a]))
                nx.append(remap[int(sup.nextstate[a])])
        arc_begin.append(len(il))
    final = np.full(len(keep), np.inf, np.float32)
    final[e0 - b0:] = 0.0 if hi < S else sup.final[e0:e1]
    return type(sup)(sup.weight, hi - lo, T, sup.label_dim, len(keep), np.asarray(arc_begin, np.int32),
                     np.asarray(il, np.int32), np.asarray(aw, np.float32), np.asarray(nx, np.int32), final)


def all_reduce_results(results, group=None, device=None, even_if_alone=False):
    """SUM-all-reduces ``[objf, l2_term, weight, xent_objf]`` over the process group -- ONE collective of four
    float64 -- updates ``results`` in place and returns it.  With the ``nccl`` backend (RCCL) the buffer is built on
    the device from the kernels' own outputs (``results._dev`` / ``results._xent_dev``: no host-to-device copy) and
    one 32-byte device-to-host copy delivers the reduced values.  A group of one rank is a no-op unless
    ``even_if_alone`` (which lets a 1-GPU box exercise the RCCL branch)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        _finish_host_copy(results)
        return results
    if dist.get_world_size(group) == 1 and not even_if_alone:
        _finish_host_copy(results)
        return results
    has_xent = results._xent_dev is not None or results._xent_host is not None
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        buf = torch.zeros(4, dtype=torch.float64, device=dev)
        if results._dev is not None:
            buf[:3] = results._dev.to(dev)  # device to device
        else:
            buf[:3] = results.data.to(dev)
        if results._xent_dev is not None:
            buf[3:] = results._xent_dev.to(dev) * results._xent_scale
        elif results._xent_host is not None:
            buf[3] = results._xent_host
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        host = buf.cpu()  # the step's one device-to-host copy (32 bytes)
    else:
        _finish_host_copy(results)
        host = torch.zeros(4, dtype=torch.float64)
        host[:3] = results.data.double()
        if has_xent:
            host[3] = results.xent_objf
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    results._host.copy_(host[:3].float())
    results._stale = False
    results._dev = None  # (the device copy holds this rank's share only)
    results._defer_host_copy = False
    if has_xent:
        results.xent_objf = float(host[3])
    return results


def sync_den_graph_variant(den_graph, device, group=None):
    """Every rank of the group runs the denominator kernel rank 0 chose for ``den_graph`` (``io.DenominatorGraph.prepare``:
    from the cache of an earlier run, or timed once on rank 0's GPU).  The graph's two kernels differ in the last bits,
    so a choice made per rank -- by a timing race on each rank's GPU -- would make ranks disagree on them; the reference's
    call is deterministic for fixed inputs (``src/my_lib_chain.cpp:129-131``).  One broadcast of one integer, once per
    graph and device; ``chain_loss_data_parallel`` calls it on a graph's first use."""
    import torch.distributed as dist

    dev = torch.device(device)
    done = den_graph.__dict__.setdefault("_synced", set())
    if dev.index in done:
        return
    alone = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    if alone or dist.get_rank(group) == 0:
        den_graph.prepare(dev)
    if not alone:
        on_gpu = dist.get_backend(group) == "nccl"
        choice = torch.zeros(1, dtype=torch.int32, device=dev if on_gpu else "cpu")
        if dist.get_rank(group) == 0:
            choice[0] = den_graph.tuning(dev)["two_sequence_kernel"]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(choice, src=src, group=group)
        if dist.get_rank(group) != 0:
            den_graph.prepare(dev, variant=int(choice.item()))
    done.add(dev.index)


def _finish_host_copy(results):
    if results._defer_host_copy and results._dev is not None:
        results._host.copy_(results._dev)
        results._stale = False
    results._defer_host_copy = False


def chain_loss_data_parallel(input, den_graph, supervision, l2_regularize=0.0, leaky_hmm_coefficient=1e-5,
                             xent_regularize=0.0, xent_input=None, kaldi_way=False, group=None, even_if_alone=False):
    """``chain_loss`` on this rank's shard followed by the one all-reduce.  Returns
    ``(loss, results)`` where ``results`` holds the GLOBAL ``[objf, l2_term, weight]`` (and ``xent_objf``) and
    ``loss`` is a tensor whose value is the global ``-objf/weight`` and whose backward is this rank's local
    gradient (``-deriv``, exactly as the single-GPU wrapper; DDP then averages parameter grads).  The rank's own
    results never visit the host: the kernels' device-side floats go into the collective and one copy brings the
    reduced values back."""
    if input.is_cuda and hasattr(den_graph, "prepare"):
        sync_den_graph_variant(den_graph, input.device, group)
    results = ChainResults()
    results._defer_host_copy = bool(input.is_cuda)
    loss, results = _chain_loss_into(results, input, den_graph, supervision, l2_regularize, leaky_hmm_coefficient,
                                     xent_regularize, xent_input, kaldi_way)
    all_reduce_results(results, group=group, device=input.device if input.is_cuda else None, even_if_alone=even_if_alone)
    with torch.no_grad():
        loss.copy_(results.loss)
    return loss, results


def combine_results(list_of_results):
    """Host-side sum of per-shard ChainResults (what the all-reduce computes); for tests and logs."""
    out = ChainResults()
    xe = [r.xent_objf for r in list_of_results]
    for r in list_of_results:
        out.data += r.data
    if all(v is not None for v in xe) and xe:
        out.xent_objf = float(sum(xe))
    return out
