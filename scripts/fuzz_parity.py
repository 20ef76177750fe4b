"""Randomised parity sweep (development aid, run on the GPU box): random graph families / sizes / options
through tc_chain_objf_and_deriv against the CPU oracle.  Prints one line per case; exits non-zero on the
first mismatch beyond 1e-4 relative."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from helpers import hip_chain, rel_err  # noqa: E402
from oracle import pyoracle  # noqa: E402
from torchain_amd import io, synth  # noqa: E402

pyoracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
REL = 1e-4
for case in range(n):
    kind = rng.choice(["tied", "nearly", "hubs", "general", "l2r"])
    big_mode = len(sys.argv) > 3  # third argument: the large-layout instantiations (JV = 4, PV = 2 / 3, streamed)
    H = int(rng.choice([8192, 9000, 12000, 16384, 17000] if big_mode else [1, 2, 5, 63, 64, 65, 200, 777, 1500, 4096, 4097, 6000]))
    P = int(rng.choice([300, 4097, 6000, 9000, 12289] if big_mode else [1, 3, 17, 64, 100, 333, 1025]))
    deg = int(rng.integers(1, 7))
    seed = int(rng.integers(0, 10000))
    if kind == "tied":
        fst = synth.random_den_fst(H, max(deg, 1), P, seed=seed)
    elif kind == "nearly":
        fst = synth.nearly_tied_den_fst(max(H, 4), max(deg, 2), P, seed=seed, fraction=float(rng.uniform(0.01, 0.4)))
    elif kind == "hubs":
        Hh = max(H, 40)
        fst = synth.skewed_tied_den_fst(Hh, Hh * int(rng.integers(3, 12)), P, seed=seed)
    elif kind == "general":
        Hh = max(H, 20)
        fst = synth.skewed_den_fst(min(Hh, 1500), min(Hh, 1500) * int(rng.integers(3, 10)), P, seed=seed)
    else:
        fst = synth.left_to_right_den_fst(P, seed=seed)
    S, T = int(rng.integers(1, 6)), int(rng.integers(1, 12))
    if fst.num_states > 2000:
        S, T = min(S, 2), min(T, 5)
    leaky = float(rng.choice([1e-5, 0.05, 0.2]))
    l2 = float(rng.choice([0.0, 1e-4]))
    force = rng.choice(["", "", "TC_FORCE_BIG", "TC_FORCE_GENERAL"])
    for k in ("TC_FORCE_BIG", "TC_FORCE_GENERAL"):
        os.environ.pop(k, None)
    if force:
        os.environ[force] = "1"
    g = pyoracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=seed + 1, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed + 2)
    ref = pyoracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky, want_xent=True)
    out = hip_chain(fst, sup, y, l2=l2, leaky=leaky, xent=True)
    res = out["results"]
    kern = io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"]
    # objf = num - den is a difference of two log-probs of size ~S*T: when the numerator covers the whole
    # (degenerate) graph it is ~0 and a relative error is meaningless, hence the floor
    e_obj = abs(res[0] - ref["objf"]) / max(abs(ref["objf"]), 0.05 * S * T)
    e_der = rel_err(out["deriv"], ref["deriv"], floor=1.0)
    e_x = rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0)
    ok = e_obj <= REL and e_der <= REL and e_x <= REL and res[2] == ref["weight"]
    print("%3d %-7s H=%-5d A=%-6d P=%-4d S=%d T=%-2d leaky=%g l2=%g %-16s kernel=%d  objf %.1e deriv %.1e xent %.1e %s"
          % (case, kind, fst.num_states, len(fst.src), fst.num_pdfs, S, T, leaky, l2, force, kern, e_obj, e_der, e_x,
             "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases ok" % n)
