"""Host-logic tests of the denominator schedule builder (no GPU): the built streams -- state
permutation, cell order, row-end masks, secondary rows and fix-up lists, per-state tables -- are
replayed on the CPU exactly as the kernels consume them (tc_den_graph_debug_walk) and compared with
the definition of the arc sums ([K] chain-denominator.cc AlphaGeneralFrame / BetaDashGeneralFrame:
sum over in-arcs of w * alpha(src) * p(pdf), sum over out-arcs of w * beta(dst) * p(pdf))."""
import numpy as np
import pytest

from torchain_amd import io, synth


def definition(fst, direction, gather, pdf_factor):
    w = np.exp(-fst.weight.astype(np.float64))
    pf = pdf_factor.astype(np.float64)[fst.ilabel - 1]
    g = gather.astype(np.float64)
    if direction == 0:
        return np.bincount(fst.dst, weights=w * g[fst.src] * pf, minlength=fst.num_states)
    return np.bincount(fst.src, weights=w * g[fst.dst] * pf, minlength=fst.num_states)


def check_graph(fst, expect_kind):
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    assert graph.stats()["tied"] == expect_kind
    rng = np.random.default_rng(1)
    gather = rng.uniform(0.1, 1.0, fst.num_states).astype(np.float32)
    pdf_factor = rng.uniform(0.5, 2.0, fst.num_pdfs).astype(np.float32)
    for direction in (0, 1):
        got = graph.debug_walk(direction, gather, pdf_factor)
        ref = definition(fst, direction, gather, pdf_factor)
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got - ref).max() <= 2e-6 * scale, (direction, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("H,deg,P", [(3, 2, 5), (64, 4, 32), (1000, 5, 400), (4096, 3, 300), (5000, 3, 300), (9000, 3, 5000),
                                     (14000, 3, 2000)])
def test_tied_owner_schedules_replay(H, deg, P):
    check_graph(synth.random_den_fst(H, deg, P, seed=H + deg), 1)


def test_tied_schedules_with_hub_states_replay():
    """Arc lists of several hundred arcs: primary rows at home, secondary rows elsewhere, folded by the owner."""
    check_graph(synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 1)
    check_graph(synth.skewed_tied_den_fst(3000, 30000, 500, seed=9, hub_fraction=0.005), 1)


@pytest.mark.parametrize("H,deg,P", [(17000, 3, 900), (20000, 3, 700), (24576, 4, 2000), (28000, 6, 2928)])
def test_plane_wise_schedules_replay(H, deg, P):
    """16385..28672 positions: a wave's stream cut into sub-streams (secondary rows, then one per plane of 4096 positions),
    16-bit positions in the cells, fix-up lists per thread and plane (den_tied_planes.hip)."""
    fst = synth.random_den_fst(H, deg, P, seed=H + deg)
    assert io.DenominatorGraph(fst, P).stats()["lds_bytes"] <= 160 * 1024
    check_graph(fst, 1)


@pytest.mark.parametrize("H,deg,P", [(29000, 3, 900), (33000, 4, 2928), (40000, 3, 4096)])
def test_split_source_schedules_replay(H, deg, P):
    """28673..40960 positions (round 6): the gather source is in LDS a half at a time, so every row is cut into the cells whose
    source lies in the first ceil(planes / 2) planes and the others; a wave's stream is [sub-streams of the first half | of the
    second], a cell's field is its position inside its half, fix-up lists per half, thread and plane."""
    fst = synth.random_den_fst(H, deg, P, seed=H + deg)
    assert io.DenominatorGraph(fst, P).stats()["lds_bytes"] <= 160 * 1024
    check_graph(fst, 1)


def test_split_source_schedules_with_hub_states_replay():
    """... with in-degrees of hundreds: secondary rows of both halves share their private slots."""
    fst = synth.phone_lm_den_fst(num_histories=3200, branching=10, seed=11)
    assert fst.num_states > 28672
    check_graph(fst, 1)


def test_plane_wise_schedules_of_phone_lm_graphs_replay():
    """R4 (24000 states, 312000 arcs, in-degrees to 200: secondary rows folded per plane) stays on chip."""
    check_graph(synth.config_den_fst("R4"), 1)


def test_left_to_right_graph_replay():
    check_graph(synth.left_to_right_den_fst(200, seed=42), io.DenominatorGraph(synth.left_to_right_den_fst(200, seed=42), 200).stats()["tied"])


def test_general_schedules_replay():
    check_graph(synth.skewed_den_fst(300, 6000, 120, seed=4), 0)


def test_config3_and_config5_graphs_replay():
    check_graph(synth.config_den_fst("C3"), 1)
    check_graph(synth.config_den_fst("C5"), 1)


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
def test_streamed_tables_replay(kernel_family, width):
    """The streamed path's lists (rows bundled four or two at a time for slabs of 16 / 32 sequences, entries in
    chunks per bundle) replayed on the host against the definition."""
    kernel_family(width)
    kernel_family("no_planes")  # (a tied graph of this size would otherwise take the plane-wise on-chip kernel)
    check_graph(synth.random_den_fst(20000, 3, 700, seed=31), 2)      # tied streamed tables
    kernel_family("force_streamed")
    check_graph(synth.skewed_den_fst(300, 6000, 120, seed=4), 2)       # general streamed tables
    check_graph(synth.nearly_tied_den_fst(500, 5, 90, seed=7), 2)      # tied streamed tables of a split graph
    check_graph(synth.skewed_tied_den_fst(400, 7000, 150, seed=8), 2)  # tied, hub states, states without self-loop


def test_forced_general_matches_tied(kernel_family):
    kernel_family("force_general")
    check_graph(synth.random_den_fst(1000, 5, 400, seed=3), 0)


def test_nearly_tied_graphs_are_split_not_demoted(kernel_family):
    """A few states entered through several pdfs: the builder splits them (exactly) and keeps the graph on
    the tied kernel; with splitting disabled the same graph takes the general path."""
    fst = synth.nearly_tied_den_fst(2000, 6, 300, seed=5)
    check_graph(fst, 1)
    fst2 = synth.nearly_tied_den_fst(64, 4, 20, seed=6, fraction=0.3)
    check_graph(fst2, 1)
    kernel_family("no_split")
    check_graph(fst, 0)


def test_arbitrary_labelings_are_not_split():
    """Random arc labels are not a chain graph: splitting would multiply the states; general path."""
    check_graph(synth.skewed_den_fst(300, 6000, 120, seed=4), 0)


def test_phone_lm_graph_near_the_state_limit_stays_on_chip():
    """13800 states with in-degrees up to 132: 888 secondary rows at the default row length would not fit the LDS next
    to 16384 positions; the builder lets the home rows grow instead of giving the graph to the streamed kernels."""
    fst = synth.config_den_fst("R2")
    stats = io.DenominatorGraph(fst, fst.num_pdfs).stats()
    assert stats["tied"] == 1 and 0 < stats["lds_bytes"] <= 160 * 1024 + 1024  # (lds_bytes is quoted for T = 256)
    # (rows = non-empty arc lists: since round 5 the builder takes the row cut that leaves the fewest cells -- here one
    # without any secondary row, so the forward rows are the states that have in-arcs at all)
    has_in = len(set(int(d) for s_, d in zip(fst.src, fst.dst) if s_ != d))
    assert stats["fwd_rows"] >= has_in and stats["bwd_rows"] == fst.num_states


def test_split_graph_that_does_not_fit_goes_back_to_the_general_kernel():
    """A graph that is tied only after state splitting, whose split version (9500 work states -> 16 states per thread,
    next to 12000 pdfs) fits no owner-computes LDS layout: the ORIGINAL 5000-state graph is tried on the general
    on-chip kernel -- which it fits -- before the streamed kernels (~8x slower per arc) get it; and the schedules built
    for it replay exactly."""
    fst = synth.nearly_tied_den_fst(5000, 6, 12000, seed=3, fraction=0.9)
    check_graph(fst, 0)
    stats = io.DenominatorGraph(fst, fst.num_pdfs).stats()
    assert 0 < stats["lds_bytes"] <= 160 * 1024 and stats["fwd_rows"] >= fst.num_states


def test_schedules_are_the_same_in_every_process():
    """Every rank builds its own schedules, and the order of the kernels' float sums -- the last bits of the results -- follows them:
    the builder's searches run on fixed seeds and ordered containers, so two processes replay a graph bit for bit alike."""
    import subprocess
    import sys
    code = r'''
import hashlib
import numpy as np
from torchain_amd import io, synth
h = hashlib.sha1()
for fst in (synth.phone_lm_den_fst(num_pdfs=600, seed=3, num_histories=300, branching=8, unigram_fraction=0.05), synth.random_den_fst(5000, 4, 700, seed=9)):
    g = io.DenominatorGraph(fst, fst.num_pdfs)
    rng = np.random.default_rng(1)
    gather = rng.uniform(0.1, 1.0, fst.num_states).astype(np.float32)
    pf = rng.uniform(0.5, 2.0, fst.num_pdfs).astype(np.float32)
    for d in (0, 1):
        h.update(g.debug_walk(d, gather, pf).tobytes())
    h.update(repr(sorted(g.stats().items())).encode())
print(h.hexdigest())
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600) for _ in range(2)]
    assert all(o.returncode == 0 for o in outs), outs[0].stderr[-2000:]
    assert len(outs[0].stdout.strip()) == 40 and outs[0].stdout == outs[1].stdout
