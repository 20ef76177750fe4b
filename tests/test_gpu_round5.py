"""Round 5 ("Next round" item 3 and weak #3): the kernels for graphs beyond the 8/12/16-states-per-thread
on-chip layouts compared with the oracle AT SIZE -- R4 (24 000 states, 312 000 arcs, phone-LM structure with in-degrees
to 200: the size class of the reference recipe's own den.fst, example/chime5/train_faster.py:91) and X2 (40 000 states,
400 000 arcs) at 64 x 150 and 256 x 30, both slab widths of the streamed path (many slabs per XCD, the XCD-aware block
decode with more than 8 blocks per XCD, hub bundles with groups_sum, fixed-point L2 gamma atomics over 150 frames), with
the matrix-wise and the element-wise bounds of test_gpu_round4.py, plus the size-independent property sum(gamma) = S*T and
row sums = 1 at 256 x 150 ([K] DenominatorComputation's own check, the reference's src/chain-supervision-test.hpp:388-463).
The oracle runs its sequences in blocks on the host's cores (oracle.den_forward_backward_blocks)."""
import os

import numpy as np
import pytest

from torchain_amd import io, synth

from helpers import hip_den, rel_err
from test_gpu_round4 import elementwise

pytestmark = pytest.mark.gpu
REL = 1e-4


def _oracle_den(oracle, fst, y, S, T, leaky):
    threads = max(1, min(64, os.cpu_count() or 1))
    lp, deriv = oracle.den_forward_backward_blocks(oracle.DenGraph(fst), y, S, T, leaky, block=max(1, S // threads),
                                                   threads=threads, deriv_weight=1.0)
    return lp, deriv


def _compare(oracle, cfg, S, T, seed, expect_tied):
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    y = synth.random_nnet_output(S, T, c["P"], seed=seed)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == expect_tied, out["graph"].stats()
    assert out["status"] == 0
    ref_lp, ref = _oracle_den(oracle, fst, y, S, T, c["leaky"])
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp), (out["logprob"], ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "%s %dx%d" % (cfg, S, T))
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() <= 1e-4, np.abs(rows - 1.0).max()


# ---- the plane-wise on-chip kernel (den_tied_planes.hip): tied graphs of 16385..28672 positions --------------------------
@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_small_batches(oracle, kernel_family, form):
    """5, 6 and 7 planes, few sequences and frames, against the full objective's oracle (numerator included) -- in the form
    batches of at most half the CUs take by default (two workgroups per sequence meeting in the middle) and in the fused kernel."""
    from helpers import hip_chain
    if form == "fused":
        kernel_family("no_phase_split")
    for H, deg, P, S, T in ((17000, 3, 900, 2, 7), (24000, 4, 2000, 3, 5), (28000, 5, 2928, 2, 6)):
        fst = synth.random_den_fst(H, deg, P, seed=H)
        g = oracle.DenGraph(fst)
        sup = synth.random_supervision(fst, S, T, 2, seed=H + 1, initial_probs=g.initial_probs())
        y = synth.random_nnet_output(S, T, P, seed=H + 2)
        ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True)
        st = out["graph"].stats()
        assert st["tied"] == 1 and st["lds_bytes"] > 100 * 1024, st
        res = out["results"]
        assert abs(res[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T), (H, res, ref["results"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, H
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL, H


@pytest.mark.parametrize("S,T,form", [(64, 150, "two_cu"), (64, 150, "fused"), (256, 30, "fused"), (128, 31, "two_cu")])
def test_plane_wise_kernel_at_size(oracle, kernel_family, S, T, form):
    """R4 on its default path: on chip (tied == 1), log-prob, derivative matrix-wise and element-wise, row sums.  Batches of
    up to 128 sequences take the two-workgroup form (an odd T: the roles' halves differ), 256 the fused kernel."""
    if form == "fused":
        kernel_family("no_phase_split")
    _compare(oracle, "R4", S, T, seed=511, expect_tied=1)


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_accumulate_and_no_deriv(oracle, kernel_family, form):
    """[K] Backward(deriv_weight, &deriv) adds into deriv; the forward-only call gives the same log-prob."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.random_den_fst(20000, 3, 700, seed=31)
    S, T = 3, 9
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=41)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.05, deriv_weight=1.0)
    out = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert out["graph"].stats()["tied"] == 1 and out["status"] == 0
    assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])
    assert rel_err(out["deriv"] - 0.5, ref["deriv"]) <= REL
    out2 = hip_den(fst, y, S, leaky=0.05, want_deriv=False)
    assert abs(out2["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"])


@pytest.mark.parametrize("form", ["two_cu", "fused"])
def test_plane_wise_kernel_is_bitwise_reproducible_and_slices(kernel_family, form):
    """Sequences never interact: a 5-sequence call's rows equal the rows of the same sequences in an 8-sequence call."""
    if form == "fused":
        kernel_family("no_phase_split")
    fst = synth.config_den_fst("R4")
    P = synth.CONFIGS["R4"]["P"]
    S, T = 8, 12
    y = synth.random_nnet_output(S, T, P, seed=43)
    g = io.DenominatorGraph(fst, P)
    a = hip_den(fst, y, S, leaky=0.1, graph=g)
    b = hip_den(fst, y, S, leaky=0.1, graph=g)
    assert a["logprob"] == b["logprob"] and np.array_equal(a["deriv"], b["deriv"])
    sub = np.ascontiguousarray(y.reshape(T, S, P)[:, :5].reshape(T * 5, P))
    c = hip_den(fst, sub, 5, leaky=0.1, graph=g)
    assert np.array_equal(c["deriv"].reshape(T, 5, P), a["deriv"].reshape(T, S, P)[:, :5])


@pytest.mark.parametrize("form", ["two_cu", "fused"])
@pytest.mark.parametrize("scale,leaky", [(10.0, 1e-5), (10.0, 0.1), (20.0, 0.1)])
def test_plane_wise_kernel_peaky_outputs(oracle, kernel_family, scale, leaky, form):
    """y ~ N(0, scale^2), 150 frames, against the float64 log-semiring formulation with tests/test_gpu_peaky.py's bounds
    (absolute 1e-5 on posteriors, element-wise by magnitude class, and no further from the Kaldi-style float32 oracle than
    that oracle is from the truth) -- the plane-wise kernel keeps alpha UN-dashed in its history and beta' in L2."""
    from test_gpu_peaky import _check
    if form == "fused":
        kernel_family("no_phase_split")
    _check(oracle, synth.random_den_fst(17000, 3, 600, seed=77), 1, 150, scale, leaky)


@pytest.mark.parametrize("width", ["slab_narrow", "slab_wide"])
@pytest.mark.parametrize("cfg", ["R4", "X2"])
def test_streamed_path_at_size_64x150(oracle, kernel_family, cfg, width):
    kernel_family("force_streamed")
    kernel_family(width)
    _compare(oracle, cfg, 64, 150, seed=501, expect_tied=2)


@pytest.mark.parametrize("cfg,width", [("R4", "slab_narrow"), ("X2", "slab_wide")])
def test_streamed_path_at_size_256x30(oracle, kernel_family, cfg, width):
    """more slabs than XCDs (16 / 8 slabs of 16 / 32 sequences): every XCD walks several slabs one after the other"""
    kernel_family("force_streamed")
    kernel_family(width)
    _compare(oracle, cfg, 256, 30, seed=502, expect_tied=2)


@pytest.mark.parametrize("cfg", ["R4", "X2"])
def test_streamed_path_gamma_sums_at_256x150(kernel_family, cfg):
    """[K] BetaGeneralFrameDebug's invariant at the full batch: every frame's posteriors sum to one, so the derivative sums
    to S*T (no oracle needed; 150 frames of fixed-point L2 atomics near their range)."""
    kernel_family("force_streamed")
    c = synth.CONFIGS[cfg]
    fst = synth.config_den_fst(cfg)
    S, T = 256, 150
    y = synth.random_nnet_output(S, T, c["P"], seed=503)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["status"] == 0 and out["graph"].stats()["tied"] == 2
    assert out["deriv"].min() >= 0.0
    rows = out["deriv"].sum(axis=1, dtype=np.float64)
    assert np.abs(rows - 1.0).max() <= 1e-4, np.abs(rows - 1.0).max()
    assert abs(rows.sum() - S * T) <= 1e-5 * S * T


def test_kernel_choice_cache_below_the_python_layer(tmp_path, monkeypatch):
    """a caller of the C ABI alone -- tc_den_graph_create + tc_den_graph_prepare through ctypes, no
    io.DenominatorGraph -- gets the measured kernel choice of an earlier handle from the library's cache without any timing
    launch (both times reported as zero), and the file holds the entry under hash + device name."""
    import ctypes as C
    import json

    import torch
    from torchain_amd._lib import check, lib
    path = tmp_path / "tuning.json"
    monkeypatch.setenv("TORCHAIN_TUNING_CACHE", str(path))
    fst = synth.config_den_fst("R1")
    P = synth.CONFIGS["R1"]["P"]

    def make():
        h = C.c_void_p()
        src, dst, il = (np.ascontiguousarray(a, np.int32) for a in (fst.src, fst.dst, fst.ilabel))
        w, fin = (np.ascontiguousarray(a, np.float32) for a in (fst.weight, fst.final))
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        check(lib.tc_den_graph_create(C.byref(h), int(fst.num_states), len(src), p(src), p(dst), p(il), p(w), p(fin), int(fst.start), P),
              "tc_den_graph_create")
        return h

    def tuning(h):
        choice, a, b = C.c_int32(-1), C.c_float(-1), C.c_float(-1)
        check(lib.tc_den_graph_tuning(h, 0, C.byref(choice), C.byref(a), C.byref(b)), "tc_den_graph_tuning")
        return choice.value, a.value, b.value

    g1 = make()
    check(lib.tc_den_graph_prepare(g1, 0), "tc_den_graph_prepare")
    c1, f1, t1 = tuning(g1)
    assert f1 > 0.0 and t1 > 0.0  # really timed: nothing was cached
    key = "%016x:%s" % (int(lib.tc_den_graph_hash(g1)), torch.cuda.get_device_name(0))
    table = json.load(open(path))
    assert table[key]["two_sequence_kernel"] == c1 and table[key]["fused_ms"] == pytest.approx(f1, rel=1e-6)
    g2 = make()
    check(lib.tc_den_graph_prepare(g2, 0), "tc_den_graph_prepare")
    assert tuning(g2) == (c1, 0.0, 0.0)  # the cached choice, no timing launches
    # a choice shipped through tc_tuning_cache_put wins over nothing being cached for a third handle of another process
    assert lib.tc_tuning_cache_put(int(lib.tc_den_graph_hash(g1)), torch.cuda.get_device_name(0).encode(), 1 - c1, 1.0, 1.0) == 0
    g3 = make()
    check(lib.tc_den_graph_prepare(g3, 0), "tc_den_graph_prepare")
    assert tuning(g3)[0] == 1 - c1
    for h in (g1, g2, g3):
        lib.tc_den_graph_free(h)


def test_plane_wise_kernel_long_utterance(oracle, kernel_family):
    """400 frames (the frame sums live in LDS: round4(T + 1) floats behind the layout), both forms, log-prob and row sums."""
    fst = synth.random_den_fst(17000, 3, 500, seed=91)
    S, T = 2, 400
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=92)
    ref = oracle.den_forward_backward(oracle.DenGraph(fst), y, S, leaky=0.1, deriv_weight=1.0)
    for form in ("two_cu", "fused"):
        kernel_family("no_phase_split", 1 if form == "fused" else 0)
        out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
        assert out["status"] == 0 and out["graph"].stats()["tied"] == 1
        assert abs(out["logprob"] - ref["logprob"]) <= REL * abs(ref["logprob"]), form
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, form


@pytest.mark.parametrize("chunk", range(4))
def test_plane_wise_kernel_fuzz(oracle, chunk):
    """Seeded sweep over the plane-wise kernel's graphs: random chain-structured graphs, graphs with hub states (secondary
    rows folded per plane, or home rows cut longer), nearly chain-structured graphs the library splits into 16385..28672
    positions, phone-LM structure; 5 to 7 planes, 1 to 5 sequences (two workgroups per sequence, or the fused kernel), 1 to 12
    frames, through the full objective (tests/test_gpu_fuzz.py's bounds)."""
    from helpers import hip_chain
    from torchain_amd._lib import lib
    rng = np.random.default_rng(4200 + chunk)
    for _ in range(6):
        kind = str(rng.choice(["tied", "hubs", "nearly", "phone_lm"]))
        seed = int(rng.integers(0, 10000))
        P = int(rng.choice([64, 700, 2928, 4096]))
        if kind == "tied":
            fst = synth.random_den_fst(int(rng.integers(16500, 28600)), int(rng.integers(2, 7)), P, seed=seed)
        elif kind == "hubs":
            H = int(rng.integers(16500, 27000))
            fst = synth.skewed_tied_den_fst(H, H * int(rng.integers(3, 8)), P, seed=seed, hub_fraction=float(rng.choice([0.002, 0.01])))
        elif kind == "nearly":
            fst = synth.nearly_tied_den_fst(int(rng.integers(9000, 13000)), int(rng.integers(3, 6)), P, seed=seed, fraction=float(rng.uniform(0.3, 0.8)))
        else:
            fst = synth.phone_lm_den_fst(num_histories=int(rng.integers(1500, 2300)), branching=int(rng.integers(9, 13)), num_pdfs=max(P, 200), seed=seed)
        S, T = int(rng.integers(1, 6)), int(rng.integers(1, 13))
        leaky, l2 = float(rng.choice([1e-5, 0.05, 0.2])), float(rng.choice([0.0, 1e-4]))
        fused = bool(rng.integers(0, 2))
        lib.tc_debug_set(b"no_phase_split", 1 if fused else 0)
        try:
            g = oracle.DenGraph(fst)
            sup = synth.random_supervision(fst, S, T, 2, seed=seed + 1, initial_probs=g.initial_probs())
            y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=seed + 2, scale=float(rng.choice([1.0, 3.0])))
            ref = oracle.compute_chain_objf_and_deriv(g, sup, y, l2, leaky, want_xent=True)
            out = hip_chain(fst, sup, y, l2=l2, leaky=leaky, xent=True)
        finally:
            lib.tc_debug_set(b"no_phase_split", 0)
        st = out["graph"].stats()
        desc = "%s H=%d A=%d P=%d S=%d T=%d leaky=%g l2=%g fused=%d kernel=%d lds=%d" % (
            kind, fst.num_states, len(fst.src), fst.num_pdfs, S, T, leaky, l2, fused, st["tied"], st["lds_bytes"])
        res = out["results"]
        assert abs(res[0] - ref["objf"]) / max(abs(ref["objf"]), 0.05 * S * T) <= REL, desc
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, desc
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL, desc
        assert res[2] == ref["weight"], desc
        # (seven planes fit the LDS only beside at most ~3000 pdfs: 112 KB of gather source + exp(y) + gamma + 16 KB of row sums)
        if kind in ("tied", "phone_lm") and 16384 < fst.num_states <= 24576 and fst.num_pdfs <= 4096:
            assert st["tied"] == 1, desc  # (on chip: the plane-wise kernel)


# ---- the general on-chip kernel on owner-computes schedules (den_general_owner.hip) ---------------------------------------
@pytest.mark.parametrize("which", ["forced", "skewed", "three_planes_of_pdfs"])
def test_general_owner_kernel_against_round1_kernel_and_oracle(oracle, kernel_family, which):
    """General graphs of at most 8192 states take round 5's kernel (owner-computes schedules, 8-byte cells, two barriers per
    frame); `old_general` keeps round 1's.  Both against the oracle through the full objective; Kaldi's accumulate form; the
    forward-only call."""
    from helpers import hip_chain
    if which == "forced":
        kernel_family("force_general")
        fst = synth.random_den_fst(5000, 6, 900, seed=21)
    elif which == "skewed":
        fst = synth.skewed_den_fst(1500, 12000, 400, seed=22)
    else:
        kernel_family("force_general")
        fst = synth.random_den_fst(3000, 4, 9000, seed=23)
    S, T = 5, 17
    g = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, S, T, 2, seed=31, initial_probs=g.initial_probs())
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=32)
    ref = oracle.compute_chain_objf_and_deriv(g, sup, y, 1e-4, 0.1, want_xent=True)
    outs = []
    for old in (0, 1):
        kernel_family("old_general", old)
        out = hip_chain(fst, sup, y, l2=1e-4, leaky=0.1, xent=True)
        assert out["graph"].stats()["tied"] == 0
        res = out["results"]
        assert abs(res[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * S * T), (which, old, res, ref["results"])
        assert rel_err(out["deriv"], ref["deriv"], floor=1.0) <= REL, (which, old)
        assert rel_err(out["xent_deriv"], ref["xent_deriv"], floor=1.0) <= REL, (which, old)
        outs.append(out)
    assert not np.array_equal(outs[0]["deriv"], outs[1]["deriv"])  # (they ARE two kernels)
    kernel_family("old_general", 0)
    dref = oracle.den_forward_backward(g, y, S, leaky=0.05, deriv_weight=1.0)
    acc = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert abs(acc["logprob"] - dref["logprob"]) <= REL * abs(dref["logprob"]) and acc["status"] == 0
    assert rel_err(acc["deriv"] - 0.5, dref["deriv"]) <= REL
    again = hip_den(fst, y, S, leaky=0.05, deriv_weight=1.0, accumulate=True, init=0.5)
    assert np.array_equal(acc["deriv"], again["deriv"]) and acc["logprob"] == again["logprob"]  # bitwise reproducible
    fwd = hip_den(fst, y, S, leaky=0.05, want_deriv=False)
    assert abs(fwd["logprob"] - dref["logprob"]) <= REL * abs(dref["logprob"])


def test_general_owner_kernel_full_size_and_peaky(oracle, kernel_family):
    """The C3 graph forced onto the general kernel at 64 x 150 against the oracle (element-wise bounds), and a peaky T = 150
    sequence against the float64 formulation (tests/test_gpu_peaky.py's rule)."""
    from test_gpu_peaky import _check
    kernel_family("force_general")
    c = synth.CONFIGS["C3"]
    fst = synth.config_den_fst("C3")
    S, T = 64, 150
    y = synth.random_nnet_output(S, T, c["P"], seed=1241)
    out = hip_den(fst, y, S, leaky=c["leaky"], deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 0 and out["status"] == 0
    ref_lp, ref = _oracle_den(oracle, fst, y, S, T, c["leaky"])
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "C3 forced general")
    _check(oracle, synth.config_den_fst("C2"), 1, 150, 10.0, 0.1)


def test_plane_wise_pairs_with_a_co_tenant():
    """The plane-wise kernel's two-workgroup form pairs workgroups by ticket and hands rows over through flags in global
    memory, like den_tied_mitm.hip (tests/test_gpu_round4.py: test_paired_workgroups_with_a_co_tenant).  With large GEMMs of
    another stream keeping the GPU busy the pairs' workgroups are no longer co-resident by default: no hand-over may fail
    and the results must equal an undisturbed run's bit for bit."""
    import torch
    from test_gpu_round4 import _occupy_half_the_cus
    fst = synth.random_den_fst(17000, 3, 500, seed=61)
    S, T = 96, 40
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=62)
    graph = io.DenominatorGraph(fst, fst.num_pdfs)
    quiet = hip_den(fst, y, S, leaky=0.1, graph=graph)
    assert quiet["status"] == 0 and np.isfinite(quiet["logprob"]) and graph.stats()["tied"] == 1
    side = torch.cuda.Stream()
    for rep in range(3):
        busy = _occupy_half_the_cus(side, 40)
        out = hip_den(fst, y, S, leaky=0.1, graph=graph)
        side.synchronize()
        assert out["status"] == 0 and out["logprob"] == quiet["logprob"], rep
        assert np.array_equal(out["deriv"], quiet["deriv"]), rep
        del busy


def test_plane_wise_graph_through_chain_loss(oracle):
    """The drop-in call on a graph of the plane-wise class: chain_loss(x (B, C, T), den_graph, supervision) -> loss, results and
    loss.backward() (one tc_chain_step), with the cross-entropy regulariser, against the oracle (torchain/functions.py:62-138)."""
    import torch
    from torchain_amd.functions import chain_loss
    fst = synth.random_den_fst(18000, 3, 400, seed=71)
    B, T, P = 3, 11, fst.num_pdfs
    og = oracle.DenGraph(fst)
    sup = synth.random_supervision(fst, B, T, 2, seed=72, initial_probs=og.initial_probs())
    y = synth.random_nnet_output(B, T, P, seed=73)  # rows t * B + b
    ref = oracle.compute_chain_objf_and_deriv(og, sup, y, 5e-5, 0.1, want_xent=True)
    den = io.DenominatorGraph(fst, P)
    assert den.stats()["tied"] == 1 and den.stats()["lds_bytes"] > 100 * 1024
    x = torch.from_numpy(np.ascontiguousarray(y.reshape(T, B, P).transpose(1, 2, 0))).to("cuda:0").requires_grad_(True)
    xe = torch.zeros_like(x).requires_grad_(True)
    loss, results = chain_loss(x, den, io.Supervision.from_synth(sup), l2_regularize=5e-5, leaky_hmm_coefficient=0.1,
                               xent_regularize=0.1, xent_input=xe, kaldi_way=True)
    loss.backward()
    torch.cuda.synchronize()
    got = results.data.numpy()
    assert abs(got[0] - ref["objf"]) <= REL * max(abs(ref["objf"]), 0.05 * B * T) and got[2] == ref["weight"]
    grad = x.grad.cpu().numpy().transpose(2, 0, 1).reshape(T * B, P)
    assert rel_err(-grad, ref["deriv"], floor=1.0) <= REL
    xgrad = xe.grad.cpu().numpy().transpose(2, 0, 1).reshape(T * B, P)
    assert rel_err(-xgrad, 0.1 * ref["xent_deriv"], floor=0.1) <= REL


# ---- positions a tied layout leaves unused (schedule_owner.cpp: phantom positions carry the pdf of their lane) -----------
@pytest.mark.parametrize("H,deg,P,S,T", [(9000, 6, 1500, 5, 9), (13000, 5, 2928, 130, 4), (5000, 8, 40, 3, 8), (18000, 4, 2000, 2, 6)])
def test_unused_positions_add_nothing(oracle, kernel_family, H, deg, P, S, T):
    """A graph whose last plane of 4096 positions is mostly empty (3288 / 3384 / 3192 / 2480 unused positions; 12, 16 and 8 states
    per thread and the plane-wise kernel; one batch above 128 sequences for the fused form; 40 pdfs: fewer than lanes) gives the
    oracle's derivative, and bit for bit the derivative it gave when the unused positions all pointed at pdf 0: they add zero
    wherever they point -- what changed is that 64 lanes no longer queue on one LDS address."""
    fst = synth.random_den_fst(H, deg, P, seed=H + 5)
    y = synth.random_nnet_output(S, T, P, seed=H + 6)
    out = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert out["graph"].stats()["tied"] == 1 and out["status"] == 0
    ref_lp, ref = _oracle_den(oracle, fst, y, S, T, 0.1)
    assert abs(out["logprob"] - ref_lp) <= REL * abs(ref_lp)
    assert rel_err(out["deriv"], ref, floor=1.0) <= REL
    elementwise(out["deriv"], ref, "unused positions %d" % H)
    kernel_family("phantom_pdf0")
    old = hip_den(fst, y, S, leaky=0.1, deriv_weight=1.0)
    assert old["logprob"] == out["logprob"]
    assert np.array_equal(old["deriv"], out["deriv"])


# ---- the derivative element-wise against the float64 formulation ---------------------------------
import functools


@functools.lru_cache(maxsize=None)
def _float64_truth(cfg, S, T, leaky):
    """(graph, outputs, float64 log-prob and occupation matrix) of a workload: shared by the kernel forms that are compared with it"""
    from oracle import independent_f64 as ind
    from oracle import pyoracle
    fst = synth.config_den_fst(cfg)
    pi = pyoracle.DenGraph(fst).initial_probs()
    y = synth.random_nnet_output(S, T, fst.num_pdfs, seed=11)
    lp, gam = ind.den_logprob_and_deriv(fst, pi, np.clip(y, -30, 30), S, leaky)
    return fst, y, lp, np.asarray(gam, np.float64)


@pytest.mark.parametrize("form", ["default", "force_mitm", "no_phase_split"])
@pytest.mark.parametrize("cfg,S,T", [("C2", 2, 150), ("R1", 2, 100), ("R3", 2, 60)])
def test_derivative_elementwise_against_float64(kernel_family, cfg, S, T, form):
    """BASELINE.json's "within 1e-4 relative", read element-wise and against the float64 formulation (oracle/independent_f64.py:
    log-semiring, no scaling) instead of the Kaldi-style float32 oracle, whose own distance from it is 2-4e-6 here: on N(0, 1)
    outputs every entry above 1e-4 of the denominator's occupation matrix is within 1e-4 relative and every entry above 1e-3 within
    2e-5 (measured, all three forms, both leaky coefficients: at most 5.9e-5 / 1.02e-5, profiles/r05_gamma_accuracy.txt) -- on the
    metric's graph, on a phone-LM-structured graph and on one that reaches the tied kernels through state splitting; two CUs per
    sequence (the default at this batch, in both of its forms) and the fused kernel."""
    if form != "default":
        kernel_family(form)
    for leaky in (0.1, 1e-5):
        fst, y, lp, ref = _float64_truth(cfg, S, T, leaky)
        out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
        assert out["status"] == 0 and abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
        got = np.asarray(out["deriv"], np.float64)
        for floor, tol in ((1e-4, 1e-4), (1e-3, 2e-5)):
            m = ref > floor
            assert m.sum() > 1000
            worst = float((np.abs(got[m] - ref[m]) / ref[m]).max())
            assert worst <= tol, (cfg, form, leaky, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))


@pytest.mark.parametrize("cfg,S,T,flags", [("R4", 2, 40, ()), ("R4", 2, 40, ("no_phase_split",)), ("R4", 2, 40, ("no_planes",)),
                                           ("C2", 2, 60, ("force_general",)), ("C2", 2, 60, ("force_general", "old_general")),
                                           ("C2", 2, 60, ("force_streamed",)), ("C5", 2, 60, ()), ("C5", 2, 60, ("no_phase_split",))])
def test_derivative_elementwise_against_float64_other_kernels(kernel_family, cfg, S, T, flags):
    """The same reading for the other kernel families: the plane-wise kernel in both forms and the streamed path on the 24000-state
    graph, the two general on-chip kernels and the streamed path on the metric's graph, three planes of pdfs (C5).  Measured: at most
    5.9e-5 on entries above 1e-4 and 2.7e-5 on entries above 1e-3 (profiles/r05_gamma_accuracy.txt, last block)."""
    for f in flags:
        kernel_family(f)
    for leaky in (0.1, 1e-5):
        fst, y, lp, ref = _float64_truth(cfg, S, T, leaky)
        out = hip_den(fst, y, S, leaky=leaky, deriv_weight=1.0)
        assert out["status"] == 0 and abs(out["logprob"] - lp) <= 1e-6 * abs(lp)
        got = np.asarray(out["deriv"], np.float64)
        for floor, tol in ((1e-4, 1e-4), (1e-3, 5e-5)):
            m = ref > floor
            assert m.sum() > 1000
            worst = float((np.abs(got[m] - ref[m]) / ref[m]).max())
            assert worst <= tol, (cfg, flags, leaky, "entries above %g: worst relative error %.3g > %g" % (floor, worst, tol))
