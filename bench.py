#!/usr/bin/env python3
"""Headline benchmark: sequence-frames/s through the denominator forward-backward
(BASELINE.json: batch 256 x 150 frames x 4096 pdfs, CHiME5-like den graph H=8192 / A=65536).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- tc_den_forward_backward, i.e. [K] DenominatorComputation
Forward() + Backward() writing d/dy -- over one synthetic batch resident in HBM.  With N > 1 the
driver launches one process per GPU (torch.distributed.run); each rank owns its own 256-sequence
shard (weak scaling, no data-path collective) and every step ends with the path's only exchange, one
RCCL all-reduce of the three floats (objf, l2_term, weight).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", help="workload name in torchain_amd.synth.CONFIGS")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed full-objective / layout measurements")
    ap.add_argument("--cpu-seqs", type=int, default=256, help="sequences in the CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(fst, cfg, nseq):
    """Times the oracle (a port of Kaldi's single-threaded CPU DenominatorComputation) on a bounded
    sample of the same workload: `nseq` of the batch's sequences, all frames.  Work is exactly linear
    in the number of sequences, so sequence-frames/s transfers to the full batch."""
    from oracle import pyoracle
    from torchain_amd import synth
    pyoracle.build()
    g = pyoracle.DenGraph(fst)
    T, P = cfg["T"], cfg["P"]
    y = synth.random_nnet_output(nseq, T, P, seed=1234 + 3)
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        pyoracle.den_forward_backward(g, y, nseq, leaky=cfg["leaky"], deriv_weight=-1.0)
    dt = time.perf_counter() - t0
    out = {"value": reps * nseq * T / dt, "unit": "sequence-frames/s", "cores": 1, "kind": "port",
           "sample": "%d x (%d of the batch's %d sequences x %d frames), full graph, single thread (Kaldi's CPU "
                     "chain path is single-threaded); %.1f s" % (reps, nseq, cfg["S"], T, dt)}
    ncpu = os.cpu_count() or 1
    if ncpu > 1:
        threads = min(ncpu, 64)
        n2 = 8 * threads  # blocks of 8 sequences keep the inner loops vectorised
        y2 = synth.random_nnet_output(n2, T, P, seed=1234 + 4)
        t0 = time.perf_counter()
        pyoracle.den_forward_backward_blocks(g, y2, n2, T, cfg["leaky"], block=8, threads=threads, deriv_weight=-1.0)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": n2 * T / dt2, "cores": threads,
                            "sample": "%d sequences in blocks of 8, OpenMP over blocks; %.1f s" % (n2, dt2)}
    return out


def extras(graph, fst, cfg, S, T, P, dev):
    """Not part of the metric: the rest of the path around the benchmarked unit, timed after the timed
    region with HIP events -- the full objective (tc_chain_objf_and_deriv: denominator + numerator +
    finalisation) on a synthetic supervision, and the layout kernels that replace the reference's
    (B, C, T) <-> (T*B, C) permutes (functions.py:118-125, :112) next to the torch ops they replace."""
    import torch
    from torchain_amd import io, synth
    from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv, from2d_hip, to2d, to2d_hip

    def timeit(fn, n=10, warm=3):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    sup = synth.random_supervision(fst, S, T, 3, seed=7, initial_probs=graph.initial_probs())
    hsup = io.Supervision.from_synth(sup)
    y = torch.randn(S * T, P, device=dev)
    deriv = torch.empty_like(y)
    res = ChainResults()
    full = timeit(lambda: compute_chain_objf_and_deriv(graph, hsup, y, res.data, deriv, None, cfg.get("l2", 0.0),
                                                       cfg["leaky"], 0.0))
    x = y.view(T, S, P).permute(1, 2, 0).contiguous()  # (B, C, T)
    out = {"full_objective_ms": full, "objf_per_frame": float(res.data[0] / res.data[2]),
           "to2d_hip_ms": timeit(lambda: to2d_hip(x)), "to2d_torch_ms": timeit(lambda: to2d(x)),
           "from2d_neg_hip_ms": timeit(lambda: from2d_hip(y, (S, P, T), -1.0)),
           "from2d_neg_torch_ms": timeit(lambda: (-y).view(T, S, P).permute(1, 2, 0).contiguous())}
    return out


def main():
    args = parse()
    import torch
    from torchain_amd import io, synth
    from torchain_amd._lib import check, lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # TC_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank (used to check that path on a 1-GPU box)
    if world > 1 or os.environ.get("TC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    cfg = synth.CONFIGS[args.config]
    S, T, P = cfg["S"], cfg["T"], cfg["P"]
    if args.config == "C4":
        S = S // max(world, 1)  # strong-scaling variant: 2048 sequences split over the ranks
    fst = synth.config_den_fst(args.config)
    H, A = fst.num_states, len(fst.src)
    graph = io.DenominatorGraph(fst, P).prepare(dev)
    gstats = graph.stats()

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + 3 + rank)
    y = torch.randn(S * T, P, device=dev, generator=gen)
    deriv = torch.empty_like(y)
    nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    red = torch.zeros(3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    pending = [None]

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def step(with_reduce):
        rc = lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, cfg.get("l2", 0.0), 0,
            C.c_void_p(deriv.data_ptr()), deriv.stride(0),
            C.c_void_p(lp.data_ptr()) if with_reduce else None, C.c_void_p(st.data_ptr()) if with_reduce else None,
            C.c_void_p(ws.data_ptr()), nbytes, dev.index, C.c_void_p(stream.cuda_stream))
        check(rc, "tc_den_forward_backward")
        if dist is not None:
            # (objf, l2_term, weight): the path's one exchange, 12 bytes over xGMI.  Nothing on the GPU needs
            # its result (it feeds logging), so it is issued asynchronously and the next step's kernel runs
            # under it; the previous step's reduction is waited for first, the last one before the timer stops.
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = dist.all_reduce(red, async_op=True)

    # device warm-up (not a bench step, outside every timed region): the first few dozen launches after
    # an idle period run 10-30 % slow while the clocks ramp (profiles/r01_summary.json: trace_first8_ns)
    for _ in range(40):
        step(False)
    for _ in range(args.warmup):
        step(True)
    drain()
    torch.cuda.synchronize()
    status = int(st.item())
    logprob = float(lp.item())

    # ---- the timed region: exactly K steps, barrier + synchronize on both sides
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-launch duration of the dominant kernel, HIP events on the launch stream
    kern_ms = []
    for _ in range(min(args.steps, 20)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        rc = lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, cfg.get("l2", 0.0), 0,
            C.c_void_p(deriv.data_ptr()), deriv.stride(0), None, None, C.c_void_p(ws.data_ptr()), nbytes, dev.index,
            C.c_void_p(stream.cuda_stream))
        e1.record(stream)
        check(rc, "tc_den_forward_backward")
        e1.synchronize()
        kern_ms.append(e0.elapsed_time(e1))
    kern_ms = float(np.mean(kern_ms))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * S * T / (elapsed / args.steps)
        # algorithmic bytes per launch: read y once + write deriv once +
        # write and read every alpha' frame once, fp32
        bytes_alg = 8.0 * S * T * P + 8.0 * S * (T + 1) * (H + 1)
        peak = 8000.0  # GB/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
        achieved = bytes_alg / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "sequence-frames/sec through denominator fwd-bwd", "value": value,
            "unit": "sequence-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: batch %d x %d frames x %d pdfs per GPU, den graph H=%d A=%d, leaky %g"
                                   % (args.config, S, T, P, H, A, cfg["leaky"]),
                       "sequences_per_gpu": S, "frames": T, "pdfs": P, "den_states": H, "den_arcs": A,
                       "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic,
                         "kernel": "den_fwd_bwd_kernel (%s graph path)" % ("tied" if gstats["tied"] else "general"),
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes": bytes_alg},
            "check": {"den_logprob_per_frame": logprob / (S * T), "status": status},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fst, cfg, args.cpu_seqs)
        if world == 1 and not args.no_extras:
            out["extras"] = extras(graph, fst, cfg, S, T, P, dev)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
