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

from .functions import ChainResults, chain_loss


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
    """SUM-all-reduces ``results.data = [objf, l2_term, weight]`` over the process group in place (one
    12-byte collective) and returns ``results``.  With the ``nccl`` backend (RCCL) the three floats
    travel through a device tensor on ``device`` (default: current CUDA device).  A group of one rank is
    a no-op unless ``even_if_alone`` (which lets a 1-GPU box exercise the RCCL branch)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return results
    if dist.get_world_size(group) == 1 and not even_if_alone:
        return results
    backend = dist.get_backend(group)
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        buf = results.data.to(dev)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        results.data.copy_(buf)
        if getattr(results, "xent_objf", None) is not None:
            xe = torch.tensor([results.xent_objf], dtype=torch.float64, device=dev)
            dist.all_reduce(xe, op=dist.ReduceOp.SUM, group=group)
            results.xent_objf = float(xe.item())
    else:
        dist.all_reduce(results.data, op=dist.ReduceOp.SUM, group=group)
        if getattr(results, "xent_objf", None) is not None:
            xe = torch.tensor([results.xent_objf], dtype=torch.float64)
            dist.all_reduce(xe, op=dist.ReduceOp.SUM, group=group)
            results.xent_objf = float(xe.item())
    return results


def chain_loss_data_parallel(input, den_graph, supervision, l2_regularize=0.0, leaky_hmm_coefficient=1e-5,
                             xent_regularize=0.0, xent_input=None, kaldi_way=False, group=None, even_if_alone=False):
    """``chain_loss`` on this rank's shard followed by the one all-reduce.  Returns
    ``(loss, results)`` where ``results`` holds the GLOBAL ``[objf, l2_term, weight]`` and ``loss`` is a
    tensor whose value is the global ``-objf/weight`` and whose backward is this rank's local
    gradient (``-deriv``, exactly as the single-GPU wrapper; DDP then averages parameter grads)."""
    loss, results = chain_loss(input, den_graph, supervision, l2_regularize, leaky_hmm_coefficient,
                               xent_regularize, xent_input, kaldi_way)
    all_reduce_results(results, group=group, device=input.device if input.is_cuda else None, even_if_alone=even_if_alone)
    with torch.no_grad():
        loss.copy_(results.loss)
    return loss, results


def combine_results(list_of_results):
    """Host-side sum of per-shard ChainResults (what the all-reduce computes); for tests and logs."""
    out = ChainResults()
    for r in list_of_results:
        out.data += r.data
    return out
