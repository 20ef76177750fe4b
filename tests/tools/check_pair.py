"""Development aid (GPU box): the two-sequence kernel (den_tied_pair.hip) against the fused kernel and the oracle,
across batch shapes (odd batches, short T, hub graphs), then its time at C3."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402

from helpers import hip_den, rel_err  # noqa: E402
from oracle import pyoracle  # noqa: E402
from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402


def force(key, v):
    check(lib.tc_debug_set(key.encode(), v), key)


def compare(name, fst, S, T, leaky, seed, with_oracle=True, l2=0.0, accumulate=False):
    P = fst.num_pdfs
    y = synth.random_nnet_output(S, T, P, seed=seed)
    force("force_pair", 0); force("no_pair", 1); force("no_phase_split", 1)
    graph = io.DenominatorGraph(fst, P)
    a = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    force("no_pair", 0); force("force_pair", 1)
    t0 = time.time()
    b = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    dt = time.time() - t0
    force("force_pair", 0); force("no_phase_split", 0)
    msg = "%-28s S=%-4d T=%-4d pair vs fused: logprob %.3e  deriv %.3e  status %d/%d" % (
        name, S, T, abs(a["logprob"] - b["logprob"]) / abs(a["logprob"]), rel_err(b["deriv"], a["deriv"]), a["status"], b["status"])
    if with_oracle:
        ref = pyoracle.den_forward_backward(pyoracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
        d = ref["deriv"] - l2 * y + (0.25 if accumulate else 0.0)
        msg += " | vs oracle: logprob %.3e deriv %.3e (fused %.3e)" % (
            abs(b["logprob"] - ref["logprob"]) / abs(ref["logprob"]), rel_err(b["deriv"], d), rel_err(a["deriv"], d))
    print(msg + "  [%.2fs]" % dt, flush=True)


pyoracle.build()
small = synth.random_den_fst(256, 6, 100, seed=5)
compare("small graph", small, 4, 20, 0.1, 1)
compare("small graph odd S", small, 5, 7, 1e-5, 2)
compare("small graph T=2", small, 2, 2, 0.1, 3)
compare("small graph T=3 S=1", small, 1, 3, 0.1, 4)
compare("small graph accumulate+l2", small, 6, 11, 0.1, 5, l2=5e-5, accumulate=True)
mid = synth.random_den_fst(3000, 8, 1500, seed=6)
compare("3000 states", mid, 7, 30, 0.1, 6)
if hasattr(synth, "phone_lm_den_fst"):
    r1 = synth.config_den_fst("R1")
    compare("R1 (hub states)", r1, 6, 20, 0.1, 7)
c3 = synth.config_den_fst("C3")
compare("C3 graph", c3, 16, 150, 0.1, 8)
compare("C3 graph leaky 1e-5", c3, 9, 150, 1e-5, 9)
compare("C3 graph 300 seq", c3, 300, 40, 0.1, 10, with_oracle=False)
