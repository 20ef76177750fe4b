"""Wall time of the Python-level drop-in call (chain_loss forward + backward) at a BASELINE config (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd.functions import chain_loss  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
layout3d = len(sys.argv) > 2 and sys.argv[2] == "3d"
cfg = synth.CONFIGS[name]
S, T, P = cfg["S"], cfg["T"], cfg["P"]
fst = synth.config_den_fst(name)
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P)
sup = io.Supervision.from_synth(synth.random_supervision(fst, S, T, 3, seed=7, initial_probs=graph.initial_probs()))
x = torch.randn(S, P, T, device=dev) if layout3d else torch.randn(S * T, P, device=dev)
x.requires_grad_(True)
# XENT=1: with the cross-entropy regulariser of the reference recipe (xent_regularize 0.1, a second output of the net)
xent = os.environ.get("XENT", "") not in ("", "0")
kaldi_way = os.environ.get("KALDI_WAY", "1") not in ("", "0")
xe = torch.randn_like(x).requires_grad_(True) if xent else None


def step():
    loss, res = chain_loss(x, graph, sup, l2_regularize=cfg.get("l2", 0.0), leaky_hmm_coefficient=cfg["leaky"],
                           xent_regularize=0.1 if xent else 0.0, xent_input=xe, kaldi_way=kaldi_way)
    # (the gradient as the model's backward would receive it: loss.backward() into a leaf adds a 629 MB copy or
    # accumulate pass per step that is not part of the path)
    torch.autograd.grad(loss, [x, xe] if xent else x)
    return res


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    step()
e1.record()
e1.synchronize()
print("%s %s: chain_loss + backward  wall %.3f ms/step, device (events) %.3f ms/step" % (
    name, "(B,C,T)" if layout3d else "(T*B,C)", wall, e0.elapsed_time(e1) / n))
