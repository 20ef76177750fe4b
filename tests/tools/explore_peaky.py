"""Development aid (runs on the GPU box): HIP kernels and the fp32 oracle against the float64 formulation on peaky
outputs; the committed output is profiles/r02_peaky.txt."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import hip_den
from oracle import pyoracle, independent_f64 as ind
from torchain_amd import synth
from torchain_amd._lib import lib
pyoracle.build()
def elem(got, ref, lo):
    m = ref > lo
    return (float((np.abs(got[m] - ref[m]) / ref[m]).max()), int(m.sum())) if m.any() else (0.0, 0)
def run(name, fst, S, T, scale, leaky, force=None):
    if force: lib.tc_debug_set(force.encode(), 1)
    g = pyoracle.DenGraph(fst)
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=11, scale=scale)
    ref = pyoracle.den_forward_backward(g, y, S, leaky=leaky, deriv_weight=1.0)
    lp, gam = ind.den_logprob_and_deriv(fst, g.initial_probs(), np.clip(y, -30, 30), S, leaky)
    out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
    if force: lib.tc_debug_set(force.encode(), 0)
    e6, n6 = elem(out["deriv"], gam, 1e-6); e4, n4 = elem(out["deriv"], gam, 1e-4); e2, n2 = elem(out["deriv"], gam, 1e-2)
    o6, _ = elem(ref["deriv"], gam, 1e-6); o4, _ = elem(ref["deriv"], gam, 1e-4); o2, _ = elem(ref["deriv"], gam, 1e-2)
    print("%-22s scale=%-3g leaky=%-6g hip-f64 max-abs %.1e (oracle-f64 %.1e) lp rel %.1e (oracle %.1e)  elem>1e-6 %.1e/%.1e (%d) >1e-4 %.1e/%.1e >1e-2 %.1e/%.1e"
          % (name, scale, leaky, np.abs(out["deriv"] - gam).max(), np.abs(ref["deriv"] - gam).max(), abs(out["logprob"] - lp) / abs(lp), abs(ref["logprob"] - lp) / abs(lp),
             e6, o6, n6, e4, o4, e2, o2), flush=True)
c2 = synth.config_den_fst("C2")
for scale in (1, 5, 10, 20):
    for leaky in (1e-5, 0.1):
        run("C2 tied S=1 T=150", c2, 1, 150, scale, leaky)
small = synth.random_den_fst(300, 5, 100, seed=32)
for scale in (5, 20):
    run("small tied", small, 2, 150, scale, 0.1)
    run("small streamed", small, 2, 150, scale, 0.1, force="force_streamed")
    run("small general", small, 2, 150, scale, 0.1, force="force_general")
