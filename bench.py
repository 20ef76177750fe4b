#!/usr/bin/env python3
"""Headline benchmark: sequence-frames/s through the denominator forward-backward
(BASELINE.json: batch 256 x 150 frames x 4096 pdfs, CHiME5-like den graph H=8192 / A=65536).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- tc_den_forward_backward, i.e. [K] DenominatorComputation
Forward() + Backward() writing d/dy, with the batch's denominator log-prob and the t=0 status reduced on
the device -- over one synthetic batch resident in HBM.  With N > 1 there is one process per GPU: launched
by the driver through torch.distributed.run, or, when this script is started plainly with --gpus N, by
this script itself as N fresh children (before anything here touches the GPU).  Each rank owns its own
256-sequence shard (weak scaling, no data-path collective) and every step ends with the path's only
exchange, one RCCL all-reduce of four float64 (objf, l2_term, weight, xent objective) of the rank's shard.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", help="workload name in torchain_amd.synth.CONFIGS")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the oracle check")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed full-objective / layout measurements")
    ap.add_argument("--cpu-seqs", type=int, default=256, help="sequences in the CPU-baseline sample")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks as children of a fresh
    torch.distributed.run (this process has not touched the GPU and never will) and exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(fst, cfg, y_np, nseq):
    """Times the oracle (a port of Kaldi's single-threaded CPU DenominatorComputation) on a bounded
    sample of the same workload: the first `nseq` of the batch's sequences, all frames.  Work is exactly
    linear in the number of sequences, so sequence-frames/s transfers to the full batch.  Returns the
    baseline record and the oracle's outputs on the sample (the checker of the GPU run)."""
    import numpy as np
    from oracle import pyoracle
    from torchain_amd import synth
    pyoracle.build()
    g = pyoracle.DenGraph(fst)
    S, T, P = cfg["S"], cfg["T"], cfg["P"]
    y = np.ascontiguousarray(y_np.reshape(T, S, P)[:, :nseq, :].reshape(T * nseq, P))
    reps = 2
    ref = None
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = pyoracle.den_forward_backward(g, y, nseq, leaky=cfg["leaky"], deriv_weight=-1.0)
    dt = time.perf_counter() - t0
    out = {"value": reps * nseq * T / dt, "unit": "sequence-frames/s", "cores": 1, "kind": "port",
           "sample": "%d x (%d of the batch's %d sequences x %d frames), full graph, single thread (Kaldi's CPU "
                     "chain path is single-threaded); %.1f s" % (reps, nseq, S, T, dt)}
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
    return out, ref


def load_secondary(config, kern_ms):
    path = os.path.join(ROOT, "profiles", "secondary_%s.json" % config)
    if not os.path.exists(path):
        return None
    try:
        sec = json.load(open(path))
        return {"lds_active_frac": sec["lds_active_ms"] / kern_ms, "valu_issue_frac": sec["valu_issue_ms"] / kern_ms,
                "lds_conflict_ratio": sec["lds_conflict_ratio"], "lds_active_ms": sec["lds_active_ms"],
                "valu_issue_ms": sec["valu_issue_ms"], "clock_mhz": sec["clock_mhz"], "kernel": sec.get("kernel"),
                "source": "profiles/secondary_%s.json: %s" % (config, sec.get("source", ""))}
    except Exception:
        return None


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

    # BASELINE.json configs[1] (the reference recipe's batch: 64 sequences, same graph): the denominator alone, as a
    # small batch runs it (forward and backward recursion on two CUs, den_tied_split.hip) and in the fused kernel
    def den_ms(S2, fused):
        import ctypes as C
        from torchain_amd._lib import check, lib
        check(lib.tc_debug_set(b"no_phase_split", 1 if fused else 0), "tc_debug_set")
        y2, d2 = y[:S2 * T], deriv[:S2 * T]
        nb = lib.tc_chain_workspace_bytes(graph.ptr, S2, T)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        try:
            return timeit(lambda: check(lib.tc_den_forward_backward(
                graph.ptr, S2, C.c_void_p(y2.data_ptr()), S2 * T, P, y2.stride(0), cfg["leaky"], -1.0, 0.0, 0,
                C.c_void_p(d2.data_ptr()), d2.stride(0), None, None, C.c_void_p(ws.data_ptr()), nb, dev.index or 0,
                C.c_void_p(st)), "den"), n=20, warm=20)
        finally:
            lib.tc_debug_set(b"no_phase_split", 0)

    # the training-side step as the reference's recipe makes it: chain_loss + backward on a (B, C, T) output, with and
    # without the cross-entropy regulariser (xent_regularize 0.1, a second output), kaldi_way and the reference's default
    from torchain_amd.functions import chain_loss

    def train_step(xent, kaldi_way):
        a = x.detach().clone().requires_grad_(True)
        b = torch.randn_like(a).requires_grad_(True) if xent else None

        def step():
            loss, _res = chain_loss(a, graph, hsup, cfg.get("l2", 0.0), cfg["leaky"], 0.1 if xent else 0.0, b, kaldi_way)
            torch.autograd.grad(loss, [a, b] if xent else a)
        return timeit(step, n=20, warm=8)

    # the evaluation step: the same call under torch.no_grad() (the recipe's validation loop, example/chime5/train.py:150-171):
    # forward recursions only, no gradient tensors (tc_chain_step with grad == NULL)
    def eval_step(xent):
        b = torch.randn_like(x) if xent else None

        def step():
            with torch.no_grad():
                chain_loss(x, graph, hsup, cfg.get("l2", 0.0), cfg["leaky"], 0.1 if xent else 0.0, b, True)
        return timeit(step, n=20, warm=8)

    steps = {"train_step_bct_ms": train_step(False, True), "train_step_bct_xent_kaldi_way_ms": train_step(True, True),
             "train_step_bct_xent_reference_way_ms": train_step(True, False),
             "eval_step_bct_ms": eval_step(False), "eval_step_bct_xent_ms": eval_step(True)}

    # A graph of the size a chain recipe's own den.fst has (synth R4: 24000 states, 312000 arcs, in-degrees to 200;
    # example/chime5/train_faster.py:91 hands the recipe's den.fst to src/my_lib_example.cpp:129-134): the plane-wise on-chip
    # kernel (csrc/den_tied_planes.hip) at the headline batch and, as two workgroups per sequence, at the recipe's batch.
    def big_graph_ms():
        import ctypes as C
        from torchain_amd._lib import check, lib
        c4 = synth.CONFIGS["R4"]
        f4 = synth.config_den_fst("R4")
        g4 = io.DenominatorGraph(f4, c4["P"]).prepare(dev)
        y4 = torch.randn(S * T, c4["P"], device=dev)
        d4 = torch.empty_like(y4)
        st = torch.cuda.current_stream().cuda_stream
        res4 = {"r4_kernel_family": g4.stats()["tied"]}
        for S4 in (S, 64):
            nb = lib.tc_chain_workspace_bytes(g4.ptr, S4, T)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            res4["r4_den_batch%d_ms" % S4] = timeit(lambda: check(lib.tc_den_forward_backward(
                g4.ptr, S4, C.c_void_p(y4.data_ptr()), S4 * T, c4["P"], y4.stride(0), c4["leaky"], -1.0, 0.0, 0,
                C.c_void_p(d4.data_ptr()), d4.stride(0), None, None, C.c_void_p(ws.data_ptr()), nb, dev.index or 0,
                C.c_void_p(st)), "den"), n=5, warm=5)
            if S4 == S:
                res4["r4_secondary"] = load_secondary("R4", res4["r4_den_batch%d_ms" % S4])
                bytes4 = 8.0 * S4 * T * c4["P"] + 8.0 * S4 * (T + 1) * (f4.num_states + 1)
                res4["r4_roofline_frac"] = bytes4 / (res4["r4_den_batch%d_ms" % S4] * 1e-3) / 8.0e12
            del ws
        return res4

    out = {"full_objective_ms": full, "objf_per_frame": float(res.data[0] / res.data[2]), **steps, **big_graph_ms(),
           "batch64_den_ms": den_ms(64, False), "batch64_den_fused_kernel_ms": den_ms(64, True),
           "to2d_hip_ms": timeit(lambda: to2d_hip(x)), "to2d_torch_ms": timeit(lambda: to2d(x)),
           "from2d_neg_hip_ms": timeit(lambda: from2d_hip(y, (S, P, T), -1.0)),
           "from2d_neg_torch_ms": timeit(lambda: (-y).view(T, S, P).permute(1, 2, 0).contiguous())}
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))  # before torch / HIP is touched in this process
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    if os.environ.get("TC_BENCH_LAUNCH_TEST"):
        # CPU check of the launcher alone (tests/test_bench_launcher.py): rendezvous over gloo, the same
        # 3-float SUM all-reduce, one JSON line from rank 0; no GPU, no kernel.
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([-1.0, -0.5, 38400.0])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"launch_test": True, "n_gpus": dist.get_world_size(), "weight": float(t[2])}))
        dist.destroy_process_group()
        return

    import numpy as np
    import torch
    from torchain_amd import io, parallel, synth
    from torchain_amd._lib import check, lib
    from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # TC_BENCH_FORCE_DIST=1: take the RCCL code path with a single rank (checks that path on a 1-GPU box)
    if world > 1 or os.environ.get("TC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when its communicator comes up; stdout must carry rank 0's one
        # JSON line and nothing else, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            assert dist.get_world_size() == world
            dist.all_reduce(torch.zeros(1, device=torch.device("cuda", local_rank)))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            # the banner is written through C stdio, which buffers when stdout is a pipe: without this flush it
            # would sit in libc's buffer and come out on the restored fd 1 at exit, behind the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    cfg = synth.CONFIGS[args.config]
    S, T, P = cfg["S"], cfg["T"], cfg["P"]
    if args.config == "C4":
        S = S // max(world, 1)  # strong-scaling variant: 2048 sequences split over the ranks
    fst = synth.config_den_fst(args.config)
    H, A = fst.num_states, len(fst.src)
    graph = io.DenominatorGraph(fst, P)  # (host side: tables and schedules; uploaded below, next to its first use)
    gstats = graph.stats()
    l2 = cfg.get("l2", 0.0)

    # this rank's shard of the synthetic batch: N(0, 1) outputs, generated on the host
    # so that the CPU oracle can be run on exactly the same numbers
    y_np = synth.random_nnet_output(S, T, P, seed=1234 + 3 + rank)
    y = torch.from_numpy(y_np).to(dev)
    deriv = torch.empty_like(y)
    nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    # Upload of the graph's tables; tc_den_graph_prepare also times the graph's two kernels once.  Done
    # here, behind the seconds of host-side input generation above, the device goes into the warm-up from the clocks of
    # a running job instead of from idle (the first ~10 launches after idle run 10 - 30 % slow while the clocks ramp).
    # Every rank runs the kernel rank 0 chose (cached from an earlier run, or timed once): the two kernels differ in the last bits.
    if world > 1:
        parallel.sync_den_graph_variant(graph, dev)
    else:
        graph.prepare(dev)
    tuning = graph.tuning(dev)

    def den_step():
        rc = lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, l2, 0,
            C.c_void_p(deriv.data_ptr()), deriv.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(st.data_ptr()),
            C.c_void_p(ws.data_ptr()), nbytes, dev.index, C.c_void_p(stream.cuda_stream))
        check(rc, "tc_den_forward_backward")

    # The path's one exchange.  The three floats are REAL: (objf, l2_term, weight) of this rank's shard from
    # one untimed full-objective call; every timed step all-reduces a fresh copy of them.
    exchange = None
    local3 = red = None
    if dist is not None:
        sup = synth.random_supervision(fst, S, T, 3, seed=7 + rank, initial_probs=graph.initial_probs())
        hsup = io.Supervision.from_synth(sup)
        res = ChainResults()
        compute_chain_objf_and_deriv(graph, hsup, y, res.data, deriv, None, l2, cfg["leaky"], 0.0)
        # (parallel.all_reduce_results' buffer: four float64 -- objf, l2_term, weight, xent objective -- built on the
        # device; no xent branch here, so the fourth is 0)
        local3 = torch.zeros(4, dtype=torch.float64, device=dev)
        local3[:3] = res.data.to(dev)
        red = local3.clone()
        dist.all_reduce(red)
        torch.cuda.synchronize()
        total = red.cpu()
        assert float(total[2]) == float(world * S * T), (total, world, S, T)  # reduced weight = N * S * T
        exchange = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "collectives_per_step": 1,
                    "payload": "4 x float64", "objf": float(total[0]), "l2_term": float(total[1]),
                    "weight": float(total[2]), "loss": float(-total[0] / total[2])}

    pending = [None]

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def step():
        den_step()
        if dist is not None:
            # 32 bytes over xGMI.  Nothing on the GPU needs the result (it feeds logging), so it is issued
            # asynchronously and the next step's kernel runs under it; the previous step's reduction is
            # waited for first, the last one before the timer stops.
            drain()
            red.copy_(local3)
            pending[0] = dist.all_reduce(red, async_op=True)

    # device warm-up (not a bench step, outside every timed region): the first few dozen launches after
    # an idle period run 10-30 % slow while the clocks ramp
    for _ in range(40):
        den_step()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    status = int(st.item())
    logprob = float(lp.item())
    deriv_chk = deriv[: T * S].clone() if rank == 0 else None

    # ---- the timed region: exactly K steps, barrier + synchronize on both sides
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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
        assert float(red.cpu()[2]) == float(world * S * T)

    # ---- per-launch duration of the dominant kernel, HIP events on the launch stream
    kern_ms = []
    for _ in range(min(args.steps, 20)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        rc = lib.tc_den_forward_backward(
            graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, l2, 0,
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
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_src = ("profiles/traffic_%s.json: rocprofv3 PMC (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, "
                               "separate passes) of this kernel, collected by scripts/collect_profiles.sh, not by this run"
                               % args.config)
            except Exception:
                traffic = None
        # secondary ceilings: the LDS and VALU pipes' busy time per launch from the committed SQ counters at
        # the measured clock, as fractions of THIS run's kernel time -- the context of "x % of the HBM roofline" for a
        # kernel whose phases use these pipes one after the other
        secondary = load_secondary(args.config, kern_ms)
        tied = gstats["tied"]
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
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": (("den_tied_pair_kernel (tied graph path, two sequences per workgroup)"
                                     if tuning["two_sequence_kernel"] else "den_tied_kernel (tied graph path, fused)")
                                    if tied == 1 else
                                    "streamed kernels" if tied == 2 else "den_fwd_bwd_kernel (general graph path)"),
                         # the per-graph choice between the fused and the two-sequence kernel and the two times it was
                         # made on (zeros: taken from the cache of an earlier run, or the graph has one kernel only)
                         "kernel_choice": tuning,
                         "kernel_ms": kern_ms, "algorithmic_bytes": bytes_alg, "secondary": secondary},
            "check": {"den_logprob_per_frame": logprob / (S * T), "status": status},
        }
        if exchange is not None:
            out["exchange"] = exchange
        if not args.no_cpu_baseline:  # (rank 0, at every N: the other ranks have left their timed region)
            base, ref = cpu_baseline(fst, cfg, y_np, min(args.cpu_seqs, S))
            out["cpu_baseline"] = base
            # the oracle's outputs on the sample against the timed kernel's, same y (rows of the sample's
            # sequences; derivative = -gamma - l2*y for the GPU call, -gamma for the oracle call)
            n = min(args.cpu_seqs, S)
            got = deriv_chk.view(T, S, P)[:, :n, :].reshape(T * n, P).cpu().numpy()
            want = ref["deriv"] - np.float32(l2) * y_np.reshape(T, S, P)[:, :n, :].reshape(T * n, P)
            err = float(np.abs(got - want).max())
            chk = out["check"]
            chk["oracle_sequences"] = n
            chk["deriv_max_abs_err_vs_oracle"] = err
            chk["deriv_rel_err_vs_oracle"] = err / max(float(np.abs(want).max()), 1e-30)
            if n == S:
                chk["logprob_rel_err_vs_oracle"] = abs(logprob - ref["logprob"]) / abs(ref["logprob"])
            chk["tolerance"] = 1e-4
            chk["pass"] = bool(chk["deriv_rel_err_vs_oracle"] <= 1e-4 and
                               chk.get("logprob_rel_err_vs_oracle", 0.0) <= 1e-4 and status == 0)
        if world == 1 and not args.no_extras:
            out["extras"] = extras(graph, fst, cfg, S, T, P, dev)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
