"""Development aid (runs on the GPU box): element-wise distance of the denominator derivative from the float64 formulation
(oracle/independent_f64.py), by kernel form and by the role that wrote the frames.  Output kept in profiles/r05_gamma_accuracy.txt."""
import os, sys
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
for cfg, S, T in (("C2", 2, 150), ("R1", 2, 100), ("R3", 2, 60)):
    fst = synth.config_den_fst(cfg)
    g = pyoracle.DenGraph(fst)
    for scale, leaky in ((1, 0.1), (1, 1e-5), (5, 0.1)):
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=11, scale=scale)
        lp, gam = ind.den_logprob_and_deriv(fst, g.initial_probs(), np.clip(y, -30, 30), S, leaky)
        ref = pyoracle.den_forward_backward(g, y, S, leaky=leaky, deriv_weight=1.0)
        o4, _ = elem(ref["deriv"], gam, 1e-4); o3, _ = elem(ref["deriv"], gam, 1e-3)
        print("%s %dx%d scale %g leaky %g: fp32 oracle vs float64  >1e-4 %.2e  >1e-3 %.2e" % (cfg, S, T, scale, leaky, o4, o3), flush=True)
        for form in ("default", "force_mitm", "no_phase_split"):
            if form != "default": lib.tc_debug_set(form.encode(), 1)
            out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
            if form != "default": lib.tc_debug_set(form.encode(), 0)
            e6, n6 = elem(out["deriv"], gam, 1e-6); e4, n4 = elem(out["deriv"], gam, 1e-4); e3, n3 = elem(out["deriv"], gam, 1e-3)
            print("   %-15s hip vs float64  >1e-6 %.2e (%d)  >1e-4 %.2e (%d)  >1e-3 %.2e (%d)  max-abs %.1e  lp rel %.1e"
                  % (form, e6, n6, e4, n4, e3, n3, np.abs(out["deriv"] - gam).max(), abs(out["logprob"] - lp) / abs(lp)), flush=True)
