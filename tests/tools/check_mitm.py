"""Development aid (GPU box): the meet-in-the-middle form (den_tied_mitm.hip, ``force_mitm``) against the fused kernel and
the oracle across batch shapes and kernel instantiations, then its time next to the two-pass form."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402,F401

from helpers import hip_den, rel_err  # noqa: E402
from oracle import pyoracle  # noqa: E402
from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402


def force(key, v):
    check(lib.tc_debug_set(key.encode(), v), key)


def compare(name, fst, S, T, leaky, seed, with_oracle=True, l2=0.0, accumulate=False):
    P = fst.num_pdfs
    y = synth.random_nnet_output(S, T, P, seed=seed)
    force("force_mitm", 0); force("no_phase_split", 1)
    graph = io.DenominatorGraph(fst, P)
    a = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    force("no_phase_split", 0); force("force_mitm", 1)
    b = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    c = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0, l2_scale=l2, graph=graph, accumulate=accumulate, init=0.25)
    force("force_mitm", 0)
    msg = "%-26s S=%-3d T=%-3d mitm vs fused: logprob %.2e deriv %.2e status %d/%d repro %s" % (
        name, S, T, abs(a["logprob"] - b["logprob"]) / abs(a["logprob"]), rel_err(b["deriv"], a["deriv"]), a["status"], b["status"],
        bool(np.array_equal(b["deriv"], c["deriv"]) and b["logprob"] == c["logprob"]))
    if with_oracle:
        ref = pyoracle.den_forward_backward(pyoracle.DenGraph(fst), y, S, leaky=leaky, deriv_weight=1.0)
        d = ref["deriv"] - l2 * y + (0.25 if accumulate else 0.0)
        msg += " | vs oracle: logprob %.2e deriv %.2e (fused %.2e)" % (
            abs(b["logprob"] - ref["logprob"]) / abs(ref["logprob"]), rel_err(b["deriv"], d), rel_err(a["deriv"], d))
    print(msg, flush=True)


pyoracle.build()
small = synth.random_den_fst(256, 6, 100, seed=5)
compare("small", small, 4, 20, 0.1, 1)
compare("small odd T", small, 5, 7, 1e-5, 2)
compare("small T=2", small, 2, 2, 0.1, 3)
compare("small T=3 S=1", small, 1, 3, 0.1, 4)
compare("small accumulate l2", small, 6, 11, 0.1, 5, l2=5e-5, accumulate=True)
compare("3000 states", synth.random_den_fst(3000, 8, 1500, seed=6), 7, 30, 0.1, 6)
compare("R1 (hub states)", synth.config_den_fst("R1"), 6, 20, 0.1, 7)
compare("C3 graph T=150", synth.config_den_fst("C3"), 8, 150, 0.1, 8)
compare("C3 graph leaky 1e-5", synth.config_den_fst("C3"), 5, 150, 1e-5, 9)
compare("C5 (10240 pdfs)", synth.config_den_fst("C5"), 4, 40, 0.1, 10)
compare("P=6000 (PV=2)", synth.random_den_fst(4096, 6, 6000, seed=11), 3, 25, 0.1, 11)
compare("R3 (12 per thread)", synth.config_den_fst("R3"), 3, 20, 0.1, 12)
compare("X1 (16 per thread)", synth.config_den_fst("X1"), 2, 12, 0.1, 13, with_oracle=False)
compare("C3 graph 128 seq", synth.config_den_fst("C3"), 128, 30, 0.1, 14, with_oracle=False)
