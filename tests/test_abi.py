"""The C-ABI library: loads without a GPU, exports every symbol include/torchain_hip.h declares,
and its host-only entry points (handles, accessors, validation, den.fst reader) behave.  No compute
calls here (those need a GPU: tests/test_gpu_parity.py)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from fixtures import write_openfst_vector
from torchain_amd import io, synth
from torchain_amd._lib import LIB_PATH, TorchainHipError, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "torchain_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 20, syms
    raw = C.CDLL(LIB_PATH)
    missing = [s for s in syms if not hasattr(raw, s)]
    assert not missing, missing


def test_header_cites_the_reference_interface():
    text = open(HEADER).read()
    for ref in ("src/my_lib.h:33-42", "src/my_lib.h:29", "src/my_lib.h:21", "src/my_lib.h:23-25",
                "src/my_lib_chain.cpp:104-136"):
        assert ref in text, ref


def test_version_and_strerror():
    assert lib.tc_version() >= 100
    assert lib.tc_strerror(0) == b"ok"
    assert b"FST" in lib.tc_strerror(-2)


def test_den_graph_handle_and_initial_probs(oracle):
    fst = synth.random_den_fst(50, 4, 30, seed=1)
    g = io.DenominatorGraph(fst, fst.num_pdfs)
    assert (g.num_states, g.num_arcs, g.n_pdf) == (50, 200, 30)
    ref = oracle.DenGraph(fst).initial_probs()
    np.testing.assert_allclose(g.initial_probs(), ref, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(g.initial_probs(), synth.initial_probs_f64(fst), rtol=1e-5, atol=1e-9)
    st = g.stats()
    # tied graph: the special self-loops do not travel in the schedules, so a state whose only in-arc is
    # its self-loop has no row
    assert st["tied"] == 1 and 40 <= st["fwd_rows"] <= 50 and st["bwd_rows"] == 50
    assert st["threads"] == 1024 and 0 < st["lds_bytes"] <= 160 * 1024


def test_den_graph_schedule_covers_every_arc_once():
    """Row splitting on a skewed graph: padded slots >= arcs, more rows than states."""
    fst = synth.skewed_den_fst(300, 6000, 120, seed=4)
    st = io.DenominatorGraph(fst, fst.num_pdfs).stats()
    assert st["fwd_slots"] >= 6000 and st["bwd_slots"] >= 6000
    assert st["fwd_rows"] > 300 or st["bwd_rows"] > 300


def test_den_graph_rejects_bad_fst():
    fst = synth.random_den_fst(10, 2, 5, seed=1)
    bad = fst._replace(ilabel=fst.ilabel + 100)  # pdf out of range ([K] KALDI_ASSERT)
    with pytest.raises(TorchainHipError) as e:
        io.DenominatorGraph(bad, 5)
    assert e.value.code == -2
    bad = fst._replace(src=fst.src[::-1].copy())  # not state-major
    with pytest.raises(TorchainHipError):
        io.DenominatorGraph(bad, 5)
    bad = fst._replace(dst=(fst.dst + 100).astype(np.int32))
    with pytest.raises(TorchainHipError):
        io.DenominatorGraph(bad, 5)


def test_den_fst_file_reader(tmp_path):
    """io.DenominatorGraph(path, n_pdf) as in the reference (io.py:51-54 -> my_lib_example.cpp:129-134)."""
    fst = synth.skewed_den_fst(40, 300, 25, seed=2)
    path = str(tmp_path / "den.fst")
    write_openfst_vector(path, fst)
    g_file = io.DenominatorGraph(path, fst.num_pdfs)
    g_mem = io.DenominatorGraph(fst, fst.num_pdfs)
    assert g_file.num_states == 40 and g_file.num_arcs == 300
    np.testing.assert_array_equal(g_file.initial_probs(), g_mem.initial_probs())
    with pytest.raises(TorchainHipError) as e:
        io.DenominatorGraph(str(tmp_path / "missing.fst"), 25)
    assert e.value.code == -6
    open(str(tmp_path / "junk.fst"), "wb").write(b"not an fst at all")
    with pytest.raises(TorchainHipError):
        io.DenominatorGraph(str(tmp_path / "junk.fst"), 25)


def test_den_fst_reader_variants(tmp_path):
    """What a den.fst from a Kaldi recipe may look like besides the plain case: symbol tables attached
    (fstcompile --isymbols --keep_isymbols), non-final states, and what must be refused: a ConstFst (Kaldi's
    ReadFstKaldi goes through VectorFst::Read, which refuses it too), a non-standard arc type, an old version,
    files cut anywhere."""
    fst = synth.skewed_den_fst(30, 200, 12, seed=5)
    base = io.DenominatorGraph(fst, 12)
    p = str(tmp_path / "sym.fst")
    write_openfst_vector(p, fst, with_symbols=True)
    g = io.DenominatorGraph(p, 12)
    assert (g.num_states, g.num_arcs) == (30, 200)
    np.testing.assert_array_equal(g.initial_probs(), base.initial_probs())
    for kw in (dict(fst_type=b"const"), dict(arc_type=b"log"), dict(version=1)):
        write_openfst_vector(p, fst, **kw)
        with pytest.raises(TorchainHipError) as e:
            io.DenominatorGraph(p, 12)
        assert e.value.code == -6, kw
    n = write_openfst_vector(p, fst)
    for cut in (3, 20, 60, n // 2, n - 1):
        write_openfst_vector(p, fst, truncate_to=cut)
        with pytest.raises(TorchainHipError):
            io.DenominatorGraph(p, 12)
    write_openfst_vector(p, fst)
    with pytest.raises(TorchainHipError) as e:  # labels beyond the declared number of pdfs
        io.DenominatorGraph(p, 5)
    assert e.value.code == -2


def test_den_fst_piped_rxfilename(tmp_path):
    """Kaldi rxfilenames may be commands ("gunzip -c den.fst.gz |"): the reader runs them and parses
    their output; a failing command is an IO error."""
    import gzip
    fst = synth.random_den_fst(50, 4, 30, seed=2)
    path = tmp_path / "den.fst"
    write_openfst_vector(str(path), fst)
    with open(path, "rb") as f, gzip.open(str(path) + ".gz", "wb") as g:
        g.write(f.read())
    a = io.DenominatorGraph(str(path), 30)
    b = io.DenominatorGraph("gunzip -c %s.gz |" % path, 30)
    np.testing.assert_array_equal(a.initial_probs(), b.initial_probs())
    with pytest.raises(Exception):
        io.DenominatorGraph("false |", 30)


def test_supervision_handle_accessors():
    fst = synth.random_den_fst(30, 3, 20, seed=3)
    sup = synth.random_supervision(fst, 4, 9, 3, seed=1, weight=0.5)
    h = io.Supervision.from_synth(sup)
    assert (h.n_pdf, h.n_batch, h.n_frame) == (20, 4, 9)
    assert h.shape == (4, 9, 20) and h.weight == 0.5
    with pytest.raises(ValueError):
        io.Supervision(C.c_void_p())  # null handle, reference io.py:23-24


def test_supervision_rejects_malformed_fst():
    fst = synth.random_den_fst(30, 3, 20, seed=3)
    sup = synth.random_supervision(fst, 3, 6, 2, seed=2)
    with pytest.raises(TorchainHipError) as e:  # wrong number of frames
        io.Supervision.from_fst(1.0, 3, 7, 20, sup.arc_begin, sup.ilabel, sup.arc_weight, sup.nextstate, sup.final)
    assert e.value.code == -2
    il = sup.ilabel.copy()
    il[0] = 0  # epsilon
    with pytest.raises(TorchainHipError):
        io.Supervision.from_fst(1.0, 3, 6, 20, sup.arc_begin, il, sup.arc_weight, sup.nextstate, sup.final)


def test_supervision_not_separable_is_reported():
    """A merged FST whose boundary states do not carry proportional copies of the next sequence's
    start arcs does not factor per sequence."""
    # two sequences of one frame each; the two boundary states have different out-arc labels
    arc_begin = np.array([0, 2, 3, 4, 4], np.int32)
    ilabel = np.array([1, 2, 1, 2], np.int32)
    nxt = np.array([1, 2, 3, 3], np.int32)
    w = np.zeros(4, np.float32)
    final = np.array([np.inf, np.inf, np.inf, 0.0], np.float32)
    with pytest.raises(TorchainHipError) as e:
        io.Supervision.from_fst(1.0, 2, 1, 3, arc_begin, ilabel, w, nxt, final)
    assert e.value.code == -7


def test_den_graph_rejects_dead_states():
    """[K] asserts tot_prob > 0 per state: a state with no arcs and an infinite final weight would make the
    initial probabilities NaN."""
    fst = synth.random_den_fst(10, 3, 8, seed=1)
    keep = fst.src != 4
    final = fst.final.copy()
    final[4] = np.inf
    dead = fst._replace(src=fst.src[keep], dst=fst.dst[keep], ilabel=fst.ilabel[keep], weight=fst.weight[keep], final=final)
    with pytest.raises(TorchainHipError) as e:
        io.DenominatorGraph(dead, 8)
    assert e.value.code == -2


def test_supervision_with_final_weights_is_accepted_and_perturbation_is_not():
    """Real merged supervisions carry non-zero final weights folded into the boundary states' arc copies; the
    split accepts exactly that structure and refuses a merged FST whose copies are not proportional."""
    fst = synth.random_den_fst(30, 3, 20, seed=3)
    sup = synth.random_supervision(fst, 4, 6, 3, seed=5, final_weights=True)
    h = io.Supervision.from_synth(sup)
    assert (h.n_batch, h.n_frame) == (4, 6)
    # perturb ONE copied start arc of a boundary state that has a sibling boundary state
    times = np.full(sup.num_states, -1)
    times[0] = 0
    for st in range(sup.num_states):
        for a in range(sup.arc_begin[st], sup.arc_begin[st + 1]):
            times[sup.nextstate[a]] = times[st] + 1
    boundary = [st for st in range(sup.num_states) if times[st] == 6]
    assert len(boundary) >= 2
    w = sup.arc_weight.copy()
    assert sup.arc_begin[boundary[1] + 1] - sup.arc_begin[boundary[1]] >= 2
    w[sup.arc_begin[boundary[1]]] += 0.5
    with pytest.raises(TorchainHipError) as e:
        io.Supervision.from_fst(1.0, 4, 6, 20, sup.arc_begin, sup.ilabel, w, sup.nextstate, sup.final)
    assert e.value.code == -7


def test_workspace_has_room_for_the_two_cu_form_and_switches_exist():
    """Batches of at most 128 sequences of a tied on-chip graph may run forward and backward recursion on two CUs
    (den_tied_split.hip): their workspace holds a second history.  Larger batches and other kernel families do not
    pay for it.  The launch-time diagnostic switches are known keys."""
    fst = synth.random_den_fst(500, 4, 200, seed=1)
    g = io.DenominatorGraph(fst, fst.num_pdfs)
    assert g.stats()["tied"] == 1
    T = 40
    small, large = lib.tc_chain_workspace_bytes(g.ptr, 64, T), lib.tc_chain_workspace_bytes(g.ptr, 256, T)
    hist = 4 * (T + 1) * 4096  # bytes of one history per sequence: tied layouts hold whole planes of 4096 positions
    row = 4 * 4096             # the two-sequence form (den_tied_pair.hip) keeps one more history row per sequence
    assert small >= 2 * 64 * hist and small < 2 * 64 * hist + 64 * row + (1 << 20)
    assert large >= 256 * hist and large < 256 * hist + 256 * row + (1 << 20)
    assert lib.tc_chain_workspace_bytes(g.ptr, 128, T) > 2 * 128 * hist > lib.tc_chain_workspace_bytes(g.ptr, 129, T)
    for key in (b"no_phase_split", b"no_num_overlap", b"no_pair", b"force_pair", b"no_tune", b"no_mitm", b"force_mitm"):
        assert lib.tc_debug_set(key, 1) == 0 and lib.tc_debug_set(key, 0) == 0
    assert lib.tc_debug_set(b"no_such_switch", 1) != 0


def test_debug_switches_from_the_environment():
    """TORCHAIN_HIP_DEBUG="key[=value],..." sets the tc_debug_set switches when the library is loaded; unknown keys are
    reported on stderr and ignored (the library still loads)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); from torchain_amd._lib import lib; "
            "print(lib.tc_debug_set(b'no_tune', 0), lib.tc_debug_set(b'not_a_switch', 1))" % ROOT)
    env = dict(os.environ, TORCHAIN_HIP_DEBUG="no_tune,force_streamed=0,not_a_switch")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["0", "-1"]
    assert "unknown switch 'not_a_switch'" in out.stderr


def test_graph_hash_and_fixed_kernel_choice_need_no_gpu():
    """tc_den_graph_hash keys the cache of measured kernel choices (io.DenominatorGraph.prepare); tc_den_graph_set_variant
    fixes a choice before the graph reaches a device (nothing is uploaded here)."""
    from torchain_amd import io, synth
    from torchain_amd._lib import lib
    a = io.DenominatorGraph(synth.config_den_fst("C1"), 200)
    b = io.DenominatorGraph(synth.config_den_fst("C1"), 200)
    c = io.DenominatorGraph(synth.random_den_fst(300, 4, 200, seed=3), 200)
    ha, hb, hc = (int(lib.tc_den_graph_hash(g.ptr)) for g in (a, b, c))
    assert ha == hb and ha != hc and ha != 0 and lib.tc_den_graph_hash(None) == 0
    for v in (0, 1, -1):
        assert lib.tc_den_graph_set_variant(a.ptr, 0, v) == 0
    assert lib.tc_den_graph_set_variant(a.ptr, 0, 2) < 0 and lib.tc_den_graph_set_variant(a.ptr, 0, -2) < 0
    assert lib.tc_den_graph_set_variant(None, 0, 0) < 0


def test_tuning_cache_is_the_librarys_and_reads_the_python_format(tmp_path, monkeypatch):
    """Round 5: the cache of measured kernel choices lives below the Python layer (csrc/tuning_cache.cpp,
    consulted by tc_den_graph_prepare).  Round 6: the key ends with the generation of the kernels the choice was timed on
    (":k6"), so that a cache written against another round's kernels is not applied to this one's.  Host-only part: entries written through the C ABI are valid JSON in the format
    round 4's io.py wrote, entries written by that Python code are found by the library, a damaged file means "not cached"."""
    import ctypes as C
    import json
    path = tmp_path / "sub" / "tuning.json"
    monkeypatch.setenv("TORCHAIN_TUNING_CACHE", str(path))
    got = C.c_int32(-7)
    assert lib.tc_tuning_cache_get(0x1234, b"AMD Instinct MI355X", C.byref(got)) == 0 and got.value == -7
    assert lib.tc_tuning_cache_put(0x1234, b"AMD Instinct MI355X", 1, 0.5, 0.25) == 0
    assert lib.tc_tuning_cache_put(0xABCDEF0123456789, b"AMD Instinct MI355X", 0, 0.75, 0.875) == 0
    table = json.load(open(path))
    assert table["0000000000001234:AMD Instinct MI355X:k6"] == {"fused_ms": 0.5, "two_sequence_kernel": 1, "two_sequence_ms": 0.25}
    assert table["abcdef0123456789:AMD Instinct MI355X:k6"]["two_sequence_kernel"] == 0
    assert lib.tc_tuning_cache_get(0x1234, b"AMD Instinct MI355X", C.byref(got)) == 1 and got.value == 1
    assert lib.tc_tuning_cache_get(0xABCDEF0123456789, b"AMD Instinct MI355X", C.byref(got)) == 1 and got.value == 0
    assert lib.tc_tuning_cache_get(0x1234, b"another device", C.byref(got)) == 0
    # the round-4 Python writer's format (json.dump(indent=1, sort_keys=True))
    table["00000000000000ff:dev:k6"] = {"fused_ms": 1.0, "two_sequence_kernel": 1, "two_sequence_ms": 0.5}
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    assert lib.tc_tuning_cache_get(0xFF, b"dev", C.byref(got)) == 1 and got.value == 1
    assert lib.tc_tuning_cache_get(0x1234, b"AMD Instinct MI355X", C.byref(got)) == 1 and got.value == 1
    with open(path, "w") as f:
        f.write("{ not json")
    assert lib.tc_tuning_cache_get(0x1234, b"AMD Instinct MI355X", C.byref(got)) == 0
    assert lib.tc_tuning_cache_put(0x1, b"d", 1, 0.1, 0.05) == 0 and json.load(open(path)) == {
        "0000000000000001:d:k6": {"fused_ms": pytest.approx(0.1), "two_sequence_kernel": 1, "two_sequence_ms": pytest.approx(0.05)}}
    assert lib.tc_tuning_cache_put(0x1, None, 1, 0.1, 0.05) < 0 and lib.tc_tuning_cache_put(0x1, b"d", 2, 0.1, 0.05) < 0


def test_tuning_cache_survives_odd_device_names_and_damaged_files(tmp_path, monkeypatch):
    """A device name with quotes, a backslash or a control character still gives a JSON file (those characters are replaced in
    the key, for reader and writer alike); 1500 randomly damaged files mean "not cached" or the entry, never anything else."""
    import ctypes as C
    import json
    import random
    path = tmp_path / "tuning.json"
    monkeypatch.setenv("TORCHAIN_TUNING_CACHE", str(path))
    got = C.c_int32(0)
    odd = b'dev "B"\\\n'
    assert lib.tc_tuning_cache_put(0x1234, b"devA", 1, 0.5, 0.25) == 0 and lib.tc_tuning_cache_put(0x9999, odd, 0, 0.5, 0.75) == 0
    assert set(json.load(open(path))) == {"0000000000001234:devA:k6", "0000000000009999:dev _B___:k6"}
    assert lib.tc_tuning_cache_get(0x9999, odd, C.byref(got)) == 1 and got.value == 0
    good = open(path, "rb").read()
    rng = random.Random(3)
    for _ in range(1500):
        b = bytearray(good)
        for _ in range(rng.randint(1, 6)):
            op = rng.random()
            if op < 0.6:
                b[rng.randrange(len(b))] = rng.randrange(256)
            elif op < 0.8:
                del b[rng.randrange(len(b)):]
            else:
                b.insert(rng.randrange(len(b) + 1), rng.randrange(256))
            if not b:
                b = bytearray(b"{")
        open(path, "wb").write(bytes(b))
        got.value = 7
        rc = lib.tc_tuning_cache_get(0x1234, b"devA", C.byref(got))
        assert (rc == 1 and got.value in (0, 1)) or (rc == 0 and got.value == 7)
    assert lib.tc_tuning_cache_put(0x1234, b"devA", 1, 0.5, 0.25) == 0  # on top of whatever the last damage left
    assert lib.tc_tuning_cache_get(0x1234, b"devA", C.byref(got)) == 1 and got.value == 1
    json.load(open(path))
