import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from helpers import hip_den, rel_err
from oracle import pyoracle
from torchain_amd import io, synth
from torchain_amd._lib import check, lib
pyoracle.build()
def force(k, v): check(lib.tc_debug_set(k.encode(), v), k)
for name, T, S in (("C3", 700, 3), ("C3", 1201, 2), ("R1", 500, 3)):
    fst = synth.config_den_fst(name); P = fst.num_pdfs
    y = synth.random_nnet_output(S, T, P, seed=3)
    ref = pyoracle.den_forward_backward(pyoracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    for mode in ("no_phase_split", "force_mitm", "force_pair", "no_mitm"):
        force(mode, 1)
        g = io.DenominatorGraph(fst, P)
        try:
            out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0, graph=g)
            print("%s T=%d %-15s status %d logprob %.2e deriv %.2e" % (name, T, mode, out["status"], abs(out["logprob"] - ref["logprob"]) / abs(ref["logprob"]), rel_err(out["deriv"], ref["deriv"])), flush=True)
        except Exception as e:
            print(name, T, mode, "ERROR", e, flush=True)
        force(mode, 0)
