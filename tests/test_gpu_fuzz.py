"""Seeded randomised parity sweep (round 1 ran this as a script and kept no log): random graph families (tied,
nearly tied, hub states, arbitrary labels, left-to-right), sizes across every kernel instantiation, batch
shapes, leaky / l2 values and forced kernel families through tc_chain_objf_and_deriv against the CPU oracle.
240 small cases in 10 chunks + 24 cases on the large layouts (JV = 4, PV = 2 / 3, streamed)."""
import numpy as np
import pytest

from torchain_amd import io, synth
from torchain_amd._lib import lib

from helpers import hip_chain, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


def _case(oracle, rng, big_mode):
    kind = rng.choice(["tied", "nearly", "hubs", "general", "l2r"])
    H = int(rng.choice([8192, 9000, 12000, 16384, 17000] if big_mode else
                       [1, 2, 5, 63, 64, 65, 200, 777, 1500, 4096, 4097, 6000]))
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
    scale = float(rng.choice([1.0, 1.0, 3.0]))
    force = str(rng.choice(["", "", "force_streamed", "force_general"]))
    fused = bool(rng.integers(0, 2))  # tied on-chip graphs: the fused kernel instead of the two-CU form of small batches
    # every third case also asks for the two-sequence kernel (taken where it fits: tied on-chip graphs of at most 8192
    # positions, T >= 2; odd batches get a phantom partner) -- decided from a value already drawn, so the cases of
    # earlier rounds keep their inputs
    pair = seed % 3 == 0 and not force
    mitm = seed % 3 == 1 and not force  # ... and another third the two-CU form that meets in the middle (from 2 frames on)
    wide = seed % 2 == 1                # the streamed path (forced, or taken by the large graphs): slabs of 32 / 16 sequences
    for key in ("force_streamed", "force_general"):
        lib.tc_debug_set(key.encode(), 1 if key == force else 0)
    lib.tc_debug_set(b"no_phase_split", 1 if fused else 0)
    lib.tc_debug_set(b"force_pair", 1 if pair else 0)
    lib.tc_debug_set(b"force_mitm", 1 if mitm else 0)
    lib.tc_debug_set(b"slab_wide", 1 if wide else 0)
    lib.tc_debug_set(b"slab_narrow", 0 if wide else 1)
    try:
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=seed + 1, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed + 2, scale=scale)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky, want_xent=True)
        out = hip_chain(fst, sup, y, l2=l2, leaky=leaky, xent=True)
        kern = io.DenominatorGraph(fst, fst.num_pdfs).stats()["tied"]
    finally:
        for key in ("force_streamed", "force_general", "no_phase_split", "force_pair", "force_mitm", "slab_wide", "slab_narrow"):
            lib.tc_debug_set(key.encode(), 0)
    res = out["results"]
    # objf = num - den is a difference of two log-probs of size ~S*T: when the numerator covers the whole
    # (degenerate) graph it is ~0 and a relative error is meaningless, hence the floor
    e_obj = abs(res[0] - ref["objf"]) / max(abs(ref["objf"]), 0.05 * S * T)
    e_der = rel_err(out["deriv"], ref["deriv"], floor=1.0)
    e_x = rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0)
    desc = "%s H=%d A=%d P=%d S=%d T=%d leaky=%g l2=%g scale=%g %s%s kernel=%d: objf %.1e deriv %.1e xent %.1e" % (
        kind, fst.num_states, len(fst.src), fst.num_pdfs, S, T, leaky, l2, scale, force or ("force_pair" if pair else "force_mitm" if mitm else ""),
        " fused" if fused else "", kern,
        e_obj, e_der, e_x)
    assert e_obj <= REL and e_der <= REL and e_x <= REL and res[2] == ref["weight"], desc
    return desc


@pytest.mark.parametrize("chunk", range(10))
def test_fuzz_small_layouts(oracle, chunk):
    rng = np.random.default_rng(1000 + chunk)
    for _ in range(24):
        _case(oracle, rng, False)


@pytest.mark.parametrize("chunk", range(3))
def test_fuzz_large_layouts(oracle, chunk):
    rng = np.random.default_rng(2000 + chunk)
    for _ in range(8):
        _case(oracle, rng, True)
