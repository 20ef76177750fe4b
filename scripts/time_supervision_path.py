"""Development aid (GPU box): where a fresh minibatch's time goes between the reader and the loss: Supervision.from_synth
(tc_supervision_create, host), the first chain_loss with it (upload), later calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd.functions import chain_loss  # noqa: E402

cfg = synth.CONFIGS["C2"]
S, T, P = 64, cfg["T"], cfg["P"]
fst = synth.config_den_fst("C2")
graph = io.DenominatorGraph(fst, P)
pi = graph.initial_probs()
x = torch.randn(S * T, P, device="cuda:0", requires_grad=True)
sups = [synth.random_supervision(fst, S, T, 3, seed=s, initial_probs=pi) for s in range(12)]


def step(sup):
    loss, _ = chain_loss(x, graph, sup, l2_regularize=cfg.get("l2", 0.0), leaky_hmm_coefficient=cfg["leaky"])
    # (the gradient as the model's backward would receive it: loss.backward() into a leaf adds a 629 MB copy or
    # accumulate pass per step that is not part of the path)
    torch.autograd.grad(loss, x)


warm = io.Supervision.from_synth(sups[0])
for _ in range(10):
    step(warm)
torch.cuda.synchronize()
t_create = t_first = t_again = 0.0
for s in sups[2:]:
    t0 = time.perf_counter()
    h = io.Supervision.from_synth(s)
    t1 = time.perf_counter()
    step(h)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    step(h)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    t_create += t1 - t0
    t_first += t2 - t1
    t_again += t3 - t2
n = len(sups) - 2
print("Supervision.from_synth %.3f ms   first step with it %.3f ms   the same again %.3f ms" % (
    t_create / n * 1e3, t_first / n * 1e3, t_again / n * 1e3))
