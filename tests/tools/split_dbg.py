"""Development aid (GPU box): the split-source plane-wise kernel (graphs of 28673..40960 positions) against the oracle on tiny batches --
forward-only and forward + backward log-prob, derivative, row sums.  Test infrastructure: it imports the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import hip_den, rel_err
from oracle import pyoracle
from torchain_amd import synth
pyoracle.build()
for H, d, P, S, T in ((30000, 3, 900, 2, 1), (30000, 3, 900, 2, 2), (30000, 3, 900, 2, 5), (33000, 3, 900, 2, 3), (40000, 3, 900, 2, 3)):
    fst = synth.random_den_fst(H, d, P, seed=71)
    y = synth.random_nnet_output(S, T, P, seed=72)
    ref = pyoracle.den_forward_backward(pyoracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    fwd = hip_den(fst, y, S, leaky=0.1, want_deriv=False)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=fwd["graph"])
    print(H, T, "fwd-only logprob rel %.2e | fwd+bwd logprob rel %.2e status %d deriv rel %.2e rowsum %.4f..%.4f" % (
        abs(fwd["logprob"] - ref["logprob"]) / abs(ref["logprob"]), abs(out["logprob"] - ref["logprob"]) / abs(ref["logprob"]), out["status"],
        rel_err(out["deriv"], ref["deriv"], floor=1.0), out["deriv"].sum(1).min(), out["deriv"].sum(1).max()), flush=True)
