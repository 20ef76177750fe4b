"""Times the full chain objective (tc_chain_objf_and_deriv) on one GPU (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd.functions import ChainResults, compute_chain_objf_and_deriv  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[cfgname]
S, T, P = cfg["S"], cfg["T"], cfg["P"]
fst = synth.config_den_fst(cfgname)
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P).prepare(dev)
t0 = time.time()
sup = synth.random_supervision(fst, S, T, 3, seed=7, initial_probs=graph.initial_probs())
print("supervision generated in %.1fs: %d states %d arcs" % (time.time() - t0, sup.num_states, len(sup.ilabel)))
t0 = time.time()
hsup = io.Supervision.from_synth(sup)
print("tc_supervision_create (host split) %.3fs" % (time.time() - t0))
y = torch.randn(S * T, P, device=dev)
deriv = torch.empty_like(y)
xent = torch.empty_like(y)
res = ChainResults()
for use_xent in (False, True):
    for _ in range(30):  # past the clock ramp
        compute_chain_objf_and_deriv(graph, hsup, y, res.data, deriv, xent if use_xent else None, cfg.get("l2", 0.0),
                                     cfg["leaky"], 0.1 if use_xent else 0.0)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        compute_chain_objf_and_deriv(graph, hsup, y, res.data, deriv, xent if use_xent else None, cfg.get("l2", 0.0),
                                     cfg["leaky"], 0.1 if use_xent else 0.0)
    torch.cuda.synchronize()
    print("full objective, xent=%s: %.3f ms/call   %r" % (use_xent, (time.time() - t0) / n * 1e3, res))
