"""Development aid (GPU box): what a captured graph of the training-side step would buy.  The one-call step (tc_chain_step
through chain_loss + autograd.grad) enqueued directly vs replayed from a torch.cuda.CUDAGraph capture of the same calls:
host time per step with the GPU kept behind (enqueue cost) and wall time per step.
   python scripts/graph_probe.py C2"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd.functions import chain_loss  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = synth.CONFIGS[name]
S, T, P = cfg["S"], cfg["T"], cfg["P"]
fst = synth.config_den_fst(name)
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P).prepare(dev)
sup = io.Supervision.from_synth(synth.random_supervision(fst, S, T, 3, seed=7, initial_probs=graph.initial_probs()))
x = torch.randn(S * T, P, device=dev, requires_grad=True)


def step():
    loss, res = chain_loss(x, graph, sup, l2_regularize=cfg.get("l2", 0.0), leaky_hmm_coefficient=cfg["leaky"])
    (g,) = torch.autograd.grad(loss, x)
    return loss, g


def measure(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return host, wall


print("direct  : host %.3f ms/step, wall %.3f ms/step" % measure(step))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        step()
side.synchronize()
cg = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(cg, stream=side):
        out = step()
    print("captured: host %.3f ms/step, wall %.3f ms/step" % measure(cg.replay))
except Exception as e:  # noqa: BLE001
    print("capture failed:", repr(e)[:300])
