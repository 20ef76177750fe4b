"""The Kaldi chain-egs reader (torchain_amd/egs.py + io.Example / RandExample / open_example / print_key_length,
reference torchain/io.py:60-175 over src/my_lib_example*.cpp) on archives written by tests/kaldi_egs_writer.py and on
the committed fixture tests/golden/chain_egs.ark.  CPU only; the GPU leg is in test_gpu_parity.py."""
import io as pyio
import os

import numpy as np
import pytest
import torch

from torchain_amd import egs, io, synth

import kaldi_egs_writer as kw
from fixtures import make_example, same_fst, write_set as _write_set

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind,tol", [("FM", 0.0), ("CM2", 2e-4), ("CM3", 3e-2), ("CM", 8e-2)])
@pytest.mark.parametrize("dw", ["DW2", "DW"])
def test_example_round_trip(kind, tol, dw):
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    eg = make_example(fst, 9, seed=3, n_seq=2, final_weights=True, weight=0.5)
    blob = kw.chain_example(eg, matrix_kind=kind, dw=dw, e2e_flag=(kind == "CM2"))
    got = egs.read_chain_example(pyio.BytesIO(blob))
    assert [i["name"] for i in got["inputs"]] == ["input", "ivector"]
    for a, b in zip(got["inputs"], eg["inputs"]):
        np.testing.assert_array_equal(a["indexes"], b["indexes"])
        assert a["features"].shape == b["features"].shape
        scale = float(np.ptp(b["features"]))
        assert np.abs(a["features"] - b["features"]).max() <= tol * scale + 1e-7
    o, ref = got["outputs"][0], eg["outputs"][0]
    np.testing.assert_array_equal(o["indexes"], ref["indexes"])
    np.testing.assert_allclose(o["deriv_weights"], ref["deriv_weights"], atol=1e-6)
    s, r = o["supervision"], ref["supervision"]
    assert (s.weight, s.num_sequences, s.frames_per_sequence, s.label_dim) == (0.5, 2, 9, 24)
    assert same_fst(s, r)


def test_alignment_pdfs_block_of_later_kaldi_is_skipped(tmp_path):
    """Later Kaldi writes ``<AlignmentPdfs>`` + an integer vector behind the supervision's FST when the vector is not empty
    ([K] chain-supervision.cc: Supervision::Write); the path does not use it.  Both readers skip it (the
    native one failed on it with a misleading "</Supervision>")."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    keyed = [("a%d" % i, make_example(fst, 7, seed=30 + i, n_seq=1)) for i in range(3)]
    ark, scp = str(tmp_path / "al.ark"), str(tmp_path / "al.scp")
    kw.write_ark(ark, keyed, scp_path=scp, alignment_pdfs=True, e2e_flag=True)
    assert b"<AlignmentPdfs>" in open(ark, "rb").read()
    plain = egs.read_chain_example(pyio.BytesIO(kw.chain_example(keyed[0][1])))
    got = egs.read_chain_example(pyio.BytesIO(kw.chain_example(keyed[0][1], alignment_pdfs=True)))
    assert same_fst(got["outputs"][0]["supervision"], plain["outputs"][0]["supervision"])
    where = [(p, off) for _k, p, off in egs.read_scp(scp)]
    merged = egs.read_merged_native(where)
    ref = egs.merge_chain_examples([eg for _k, eg in keyed])
    assert same_fst(merged["outputs"][0]["supervision"], ref["outputs"][0]["supervision"])
    np.testing.assert_array_equal(merged["inputs"][0]["features"], ref["inputs"][0]["features"])


def test_end2end_flag_across_a_read_buffer_boundary(tmp_path):
    """``BufferedReader.peek(n)`` may return a single byte: with the '<' of the newer-Kaldi ``<End2End>`` token as the
    last byte of an 8192-byte buffer block a two-byte look-ahead saw only '<', left the token unread and failed with
    "bad FST magic" -- about one new-format example in 8192, ending the epoch of a sequential archive.  The key is
    padded so that the token straddles the block boundary for every alignment in turn."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    eg = make_example(fst, 9, seed=3, n_seq=2)
    blob = kw.chain_example(eg, e2e_flag=True)
    at = blob.index(b"<End2End>")
    path = str(tmp_path / "e2e.ark")
    for shift in (0, 1, 2):  # '<' at offsets 8191, 8190 and 8189 of the file
        pad = 8191 - shift - at - 2  # "key SPACE \0 B" precede the object
        key = "k" * (pad - 1)
        with open(path, "wb") as f:
            f.write(key.encode() + b" " + b"\0B" + blob)
        with open(path, "rb") as f:
            assert f.read(8192)[8191 - shift:8192 - shift] == b"<"
        for got in (list(egs.iter_archive(path)), list(egs.iter_rspecifier("ark:" + path))):  # numpy reader, native reader
            assert len(got) == 1 and got[0][0] == key
            assert same_fst(got[0][1]["outputs"][0]["supervision"], eg["outputs"][0]["supervision"])


def test_native_archive_reader_equals_the_numpy_one(tmp_path):
    """``tc_archive_*`` (sequential archives and ``command |``) against ``egs.iter_archive``: same keys, same examples."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    keyed, ark, scp = _write_set(tmp_path, fst, [5, 5, 8, 5], matrix_kind="CM")
    want = list(egs.iter_archive(ark))
    for path in (ark, "cat %s |" % ark):
        got = list(egs.iter_archive_native(path))
        assert [k for k, _ in got] == [k for k, _ in want] == [k for k, _ in keyed]
        for (_, a), (_, b) in zip(got, want):
            _same_example(a, b)
    with pytest.raises(OSError):
        list(egs.iter_archive_native(str(tmp_path / "absent.ark")))
    bad = str(tmp_path / "bad.ark")
    open(bad, "wb").write(open(ark, "rb").read()[:300])
    with pytest.raises(egs.EgsFormatError):
        list(egs.iter_archive_native(bad))


def test_malformed_examples_are_refused():
    fst = synth.random_den_fst(20, 3, 10, seed=2)
    blob = kw.chain_example(make_example(fst, 4, seed=1))
    for cut in (5, 40, len(blob) // 2, len(blob) - 3):
        with pytest.raises(egs.EgsFormatError):
            egs.read_chain_example(pyio.BytesIO(blob[:cut]))
    with pytest.raises(egs.EgsFormatError):
        egs.read_chain_example(pyio.BytesIO(blob.replace(b"<Nnet3ChainEg>", b"<Nnet3Eg>     ")))
    with pytest.raises(egs.EgsFormatError):  # a ConstFst / VectorFst where the compact acceptor must be
        egs.read_chain_example(pyio.BytesIO(blob.replace(b"compact_acceptor", b"compact_xcceptor")))


def test_append_supervisions_is_the_product_of_the_pieces(oracle):
    """[K] AppendSupervision semantics: the merged acceptor's log-partition is the sum of the pieces' and the
    posteriors are the pieces' posteriors (checked with the oracle's numerator on the merged FST), the states are in
    time order, and the merged FST is what tc_supervision_create accepts -- non-zero final weights included."""
    fst = synth.random_den_fst(30, 3, 16, seed=4)
    T = 6
    pieces = [synth.random_supervision(fst, n, T, 2, seed=10 + n, final_weights=True) for n in (1, 2, 1, 1)]
    merged = egs.append_supervisions(pieces)
    S = sum(p.num_sequences for p in pieces)
    assert (merged.num_sequences, merged.frames_per_sequence) == (S, T)
    h = io.Supervision.from_synth(merged)
    assert h.shape == (S, T, 16)
    y = synth.random_nnet_output(S, T, 16, seed=5)  # rows t*S + s
    tot = oracle.num_forward_backward(merged, y)
    acc, derivs, s0 = 0.0, np.zeros_like(y).reshape(T, S, 16), 0
    for p in pieces:
        n = p.num_sequences
        yp = np.ascontiguousarray(y.reshape(T, S, 16)[:, s0:s0 + n, :].reshape(T * n, 16))
        r = oracle.num_forward_backward(p, yp)
        acc += r["logprob_weighted"]
        derivs[:, s0:s0 + n, :] = r["deriv"].reshape(T, n, 16)
        s0 += n
    assert abs(tot["logprob_weighted"] - acc) <= 1e-5 * abs(acc)
    assert np.abs(tot["deriv"].reshape(T, S, 16) - derivs).max() <= 1e-5


def test_sequential_reader_delivers_every_example(tmp_path):
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    keyed, ark, scp = _write_set(tmp_path, fst, [5, 5, 8, 5], matrix_kind="CM")
    for rspec in ("ark:" + ark, "ark,bg:" + ark, "scp:" + scp, ark, "ark:cat %s |" % ark):
        rd = io.Example(rspec)
        seen = 0
        for (inp, aux), sup in rd:
            L = keyed[seen][1]["outputs"][0]["supervision"].frames_per_sequence
            assert sup.shape == (1, L, 24)
            assert tuple(inp.shape) == (1, 7, 3 * L + 8) and tuple(aux.shape) == (1, 3)
            assert rd.indexes.shape == (1, L) and rd.indexes[0, 1] == 3
            assert rd.deriv_weights.shape == (L,)
            seen += 1
        assert seen == 4
        with pytest.raises(ValueError):
            rd.supervision  # past the end: "null supervision ptr", as in the reference
    with io.open_example("cat " + ark) as rd:
        assert sum(1 for _ in rd) == 4


@pytest.mark.parametrize("native", [True, False])
def test_rand_reader_batches_by_length(tmp_path, native):
    """(native: the library's tc_rand_reader_* handle, the default; else the Python statement of the same reader)"""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    lengths = [5] * 7 + [8] * 4 + [11]
    keyed, ark, scp = _write_set(tmp_path, fst, lengths)
    io.print_key_length("scp:" + scp, scp + ".len")
    assert len(open(scp + ".len").read().split()) == 2 * len(lengths)
    for len_file in ("", str(tmp_path / "absent.len")):  # from the .len file / from the egs themselves
        rd = io.RandExample(scp, seed=3, batchsize=3, len_file=len_file, native=native) if len_file == "" else None
        if rd is None:
            os.rename(scp + ".len", scp + ".len.away")
            rd = io.RandExample(scp, seed=3, batchsize=3, native=native)
            os.rename(scp + ".len.away", scp + ".len")
        assert rd.n_data == 12 and rd.n_batch == 3 + 2 + 1  # ceil(7/3) + ceil(4/3) + 1
        total = 0
        for (inp, aux), sup in rd:
            B, L, P = sup.shape
            assert P == 24 and L in (5, 8, 11) and 1 <= B <= 3
            assert tuple(inp.shape) == (B, 7, 3 * L + 8) and tuple(aux.shape) == (B, 3)
            assert rd.indexes.shape == (B, L) and rd.deriv_weights.shape == (B * L,)
            total += B
        assert total == 12
    order = (lambda: [tuple(rd.batch_keys(i)) for i in range(rd.n_batch)]) if native else (lambda: [tuple(b) for b in rd._key_batch])
    first = order()
    assert sorted(k for b in first for k in b) == sorted(k for k, _ in keyed)
    rd.reset()
    assert sum(1 for _ in rd) == 6
    assert order() != first  # reshuffled


def test_rand_reader_prefetches_batches_ahead(tmp_path):
    """RandExample prepares the next minibatches on background threads while the current one is in use; the batches it
    delivers are those of the synchronous reader, in the same order, also across reset()."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    lengths = [5] * 7 + [8] * 4 + [11]
    keyed, ark, scp = _write_set(tmp_path, fst, lengths)
    io.print_key_length("scp:" + scp, scp + ".len")
    a, b = io.RandExample(scp, seed=3, batchsize=3, native=False), io.RandExample(scp, seed=3, batchsize=3, prefetch=False, native=False)
    for epoch in range(2):
        n = 0
        while a.next():
            assert b.next()
            ahead = sorted(a._pending)  # the look-ahead: the next batches are under way (or done), nothing else
            assert ahead == list(range(n + 1, min(n + 1 + a._depth, a.n_batch)))
            sa, sb = a._cur["outputs"][0]["supervision"], b._cur["outputs"][0]["supervision"]
            assert same_fst(sa, sb) and sa.num_sequences == sb.num_sequences
            np.testing.assert_array_equal(a._cur["inputs"][0]["features"], b._cur["inputs"][0]["features"])
            n += 1
        assert not b.next() and n == a.n_batch
        a.reset()
        b.reset()


def test_native_rand_reader_handle_shards_by_rank_and_looks_ahead(tmp_path):
    """``tc_rand_reader_*`` (the reference's ``my_lib_example_rand_reader_*``, src/my_lib.h:8-17): with look-ahead threads
    it delivers what it delivers without them, in the same order, also across reset(); the minibatches are those
    ``tc_example_read`` merges from the same keys; ``world`` ranks together see every batch of the epoch's list once
    (up to the ``batches % world`` left out so that all ranks take the same number of steps), and one seed gives one
    list."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    lengths = [5] * 9 + [8] * 6 + [11] * 2
    keyed, ark, scp = _write_set(tmp_path, fst, lengths)
    io.print_key_length("scp:" + scp, scp + ".len")
    where = {key: (p, off) for key, p, off in egs.read_scp(scp)}
    a, b = io.RandExample(scp, seed=5, batchsize=2, prefetch=4), io.RandExample(scp, seed=5, batchsize=2, prefetch=False)
    for epoch in range(2):
        assert a.n_batch == b.n_batch == 5 + 3 + 1 and a.n_data == 17
        keys = [a.batch_keys(i) for i in range(a.n_batch)]
        assert keys == [b.batch_keys(i) for i in range(b.n_batch)]
        n = 0
        while a.next():
            assert b.next()
            (ia, xa), sa = a.value()
            (ib, xb), sb = b.value()
            assert sa.shape == sb.shape and torch.equal(ia, ib) and torch.equal(xa, xb)
            ref = egs.read_merged_native([where[k] for k in keys[n]])
            assert same_fst(a._cur["outputs"][0]["supervision"], ref["outputs"][0]["supervision"])
            np.testing.assert_array_equal(a._cur["inputs"][0]["features"], ref["inputs"][0]["features"])
            n += 1
        assert n == a.n_batch and not b.next() and not a.next()
        with pytest.raises(ValueError):
            a.supervision  # past the end: "null supervision ptr", as in the reference
        a.reset()
        b.reset()
    # two ranks: the same list, every other batch each
    whole = io.RandExample(scp, seed=9, batchsize=2, prefetch=False)
    parts = [io.RandExample(scp, seed=9, batchsize=2, prefetch=2, rank=r, world=2) for r in range(2)]
    full = [whole.batch_keys(i) for i in range(whole.n_batch)]
    assert all(p.n_batch == len(full) // 2 for p in parts)
    for r, p in enumerate(parts):
        assert [p.batch_keys(i) for i in range(p.n_batch)] == full[r:2 * (len(full) // 2):2]
        assert sum(1 for _ in p) == p.n_batch
    with pytest.raises(ValueError):
        io.RandExample(scp, seed=9, batchsize=2, native=False, rank=1, world=2)
    with pytest.raises(egs.EgsFormatError):
        io.RandExample(scp, seed=9, batchsize=2, rank=2, world=2)


def test_reference_batch_order(tmp_path):
    """``io.RandExample(order="reference")``: the batch key lists of three seeds, two epochs each, equal those of a standalone
    restatement of ``/root/reference/src/my_lib_example_rand.cpp:119-141`` on the standard library (tests/tools/
    reference_batch_order.cpp, compiled here with g++): ``std::unordered_map<size_t, ...>`` iteration over lengths,
    ``std::shuffle`` with ``std::mt19937`` inside and across them.  Index work: exact.  Rank / world sharding applies on top."""
    import subprocess
    exe = str(tmp_path / "reference_batch_order")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "tools", "reference_batch_order.cpp")])
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    # (16 distinct lengths: the reference's hash table grows past its first bucket counts while it is filled)
    lengths = [5] * 9 + [8] * 6 + [11] * 2 + [37] * 5 + [3] * 4 + [29] * 3 + [64] + [20] * 7 + [4, 6, 7, 9, 10, 12, 13, 14, 15]
    keyed, ark, scp = _write_set(tmp_path, fst, lengths)
    # a length file in an order of its own (the reference fills its map in FILE order; io.print_key_length writes scp order)
    rng = np.random.default_rng(0)
    pairs = [(k, L) for (k, _), L in zip(keyed, lengths)]
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]
    with open(scp + ".len", "w") as f:
        f.write("".join("%s %d\n" % kv for kv in pairs))
    for seed, batchsize in ((1, 3), (12345, 2), (-7, 4)):
        want = {}
        for line in subprocess.check_output([exe, scp + ".len", str(seed), str(batchsize), "2"], text=True).splitlines():
            epoch, _, keys = line.partition(":")
            want.setdefault(int(epoch), []).append(keys.split())
        rd = io.RandExample(scp, seed=seed, batchsize=batchsize, prefetch=False, order="reference")
        parts = [io.RandExample(scp, seed=seed, batchsize=batchsize, prefetch=False, order="reference", rank=r, world=2) for r in range(2)]
        for epoch in range(2):
            got = [rd.batch_keys(i) for i in range(rd.n_batch)]
            assert got == want[epoch], (seed, epoch)
            for r, p in enumerate(parts):
                assert [p.batch_keys(i) for i in range(p.n_batch)] == want[epoch][r:2 * (len(want[epoch]) // 2):2]
                p.reset()
            rd.reset()
        assert sorted(k for b in got for k in b) == sorted(k for k, _ in keyed)
    # without a length file the reference reads the lengths from the egs in scp order (src/my_lib_example_rand.cpp:95-110): the same
    # lists as with a length file written in that order
    os.rename(scp + ".len", scp + ".len.away")
    in_scp_order = str(tmp_path / "scp_order.len")
    with open(in_scp_order, "w") as f:
        f.write("".join("%s %d\n" % (k, L) for (k, _), L in zip(keyed, lengths)))
    want0 = [line.partition(":")[2].split() for line in subprocess.check_output([exe, in_scp_order, "9", "3", "1"], text=True).splitlines()]
    rd = io.RandExample(scp, seed=9, batchsize=3, prefetch=False, order="reference")
    assert [rd.batch_keys(i) for i in range(rd.n_batch)] == want0
    os.rename(scp + ".len.away", scp + ".len")
    # the default order is another one: lengths ascending
    first, ref = io.RandExample(scp, seed=1, batchsize=3, prefetch=False), io.RandExample(scp, seed=1, batchsize=3, prefetch=False, order="reference")
    assert first.n_batch == ref.n_batch and [first.batch_keys(i) for i in range(first.n_batch)] != [ref.batch_keys(i) for i in range(ref.n_batch)]
    with pytest.raises(ValueError):
        io.RandExample(scp, seed=1, batchsize=3, order="kaldi")
    with pytest.raises(ValueError):
        io.RandExample(scp, seed=1, batchsize=3, native=False, order="reference")


def test_native_merge_equals_the_numpy_statement_and_meets_its_time_bound():
    """``tc_supervision_append`` (the library's host-side AppendSupervision; the reference merges natively,
    src/my_lib_example_rand.cpp:160) against the numpy statement of the same algorithm: identical arrays on random
    merges (several sequences per piece, final weights, pieces whose states are NOT numbered in time order); and a
    64 x 150 minibatch with ~10 arcs per frame merges in <= 5 ms (round 2's Python loops: ~1 s; the loss it feeds: ~1 ms)."""
    import time

    fst = synth.random_den_fst(400, 8, 200, seed=1)

    def identical(x, y):
        return (x.num_states == y.num_states and np.array_equal(x.arc_begin, y.arc_begin) and np.array_equal(x.ilabel, y.ilabel)
                and np.array_equal(x.nextstate, y.nextstate) and np.array_equal(x.arc_weight, y.arc_weight)
                and np.array_equal(x.final, y.final) and (x.weight, x.num_sequences, x.frames_per_sequence, x.label_dim)
                == (y.weight, y.num_sequences, y.frames_per_sequence, y.label_dim))

    def scrambled(sup, seed):
        """the same acceptor with its non-start states renumbered at random"""
        rng = np.random.default_rng(seed)
        perm = np.concatenate([[0], 1 + rng.permutation(sup.num_states - 1)])  # old -> new
        inv = np.argsort(perm)
        deg = np.diff(sup.arc_begin)[inv]
        ab = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        idx = np.concatenate([np.arange(sup.arc_begin[o], sup.arc_begin[o + 1]) for o in inv]).astype(np.int64)
        return synth.SupFst(sup.weight, sup.num_sequences, sup.frames_per_sequence, sup.label_dim, sup.num_states, ab,
                            sup.ilabel[idx], sup.arc_weight[idx], perm[sup.nextstate[idx]].astype(np.int32), sup.final[inv])

    for seed in range(8):
        pieces = [synth.random_supervision(fst, 1 + (i % 2), 12, 3, seed=seed * 10 + i, final_weights=(seed % 2 == 0))
                  for i in range(5)]
        if seed >= 4:
            pieces = [scrambled(p, seed * 7 + i) for i, p in enumerate(pieces)]
        assert identical(egs.append_supervisions(pieces), egs.append_supervisions_numpy(pieces)), seed
    # refused, not mis-merged: a piece with paths of unequal lengths
    bad = synth.random_supervision(fst, 1, 12, 3, seed=99)
    nx = bad.nextstate.copy()
    nx[0] = int(bad.nextstate[bad.arc_begin[nx[0]]])  # skip a frame
    broken = synth.SupFst(bad.weight, 1, 12, bad.label_dim, bad.num_states, bad.arc_begin, bad.ilabel, bad.arc_weight, nx, bad.final)
    with pytest.raises(egs.EgsFormatError):
        egs.append_supervisions([bad, broken])
    # refused, not read past the end of the per-time counts: a dead-end chain longer than the piece -- states
    # 0 -> 1 (final) and 0 -> 2 -> 3 -> 4 with frames = 1; the final states alone sit where they should
    inf = np.float32(np.inf)
    dead = synth.SupFst(1.0, 1, 1, bad.label_dim, 5, np.array([0, 2, 2, 3, 4, 4], np.int32), np.array([1, 2, 3, 4], np.int32),
                        np.zeros(4, np.float32), np.array([1, 2, 3, 4], np.int32), np.array([inf, 0.0, inf, inf, inf], np.float32))
    ok1 = synth.SupFst(1.0, 1, 1, bad.label_dim, 2, np.array([0, 1, 1], np.int32), np.array([1], np.int32), np.zeros(1, np.float32),
                       np.array([1], np.int32), np.array([inf, 0.0], np.float32))
    for pieces in ([dead, ok1], [ok1, dead], [ok1, dead, ok1]):
        with pytest.raises(egs.EgsFormatError):
            egs.append_supervisions(pieces)
    big = [synth.random_supervision(fst, 1, 150, 10, seed=100 + i) for i in range(64)]
    assert 9.0 <= np.mean([len(p.ilabel) / 150.0 for p in big]) <= 11.0
    egs.append_supervisions(big)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter()
        merged = egs.append_supervisions(big)
        best = min(best, time.perf_counter() - t0)
    assert merged.num_sequences == 64 and merged.frames_per_sequence == 150
    assert best <= 5e-3, "64 x 150 merge took %.2f ms" % (best * 1e3)


def test_committed_fixture(oracle):
    """tests/golden/chain_egs.ark (+ .scp, .json): written once by tests/golden/make_egs_fixture.py; the reader must
    keep giving the recorded numbers."""
    import json
    meta = json.load(open(os.path.join(HERE, "golden", "chain_egs.json")))
    got = list(egs.iter_archive(os.path.join(HERE, "golden", "chain_egs.ark")))
    assert [k for k, _ in got] == meta["keys"]
    for (key, eg), m in zip(got, meta["examples"]):
        sup = eg["outputs"][0]["supervision"]
        assert [sup.num_sequences, sup.frames_per_sequence, sup.label_dim, sup.num_states, int(sup.arc_begin[-1])] == m["sup"]
        assert abs(float(eg["inputs"][0]["features"].sum()) - m["feat_sum"]) <= 1e-3 * max(1.0, abs(m["feat_sum"]))
        assert abs(float(sup.arc_weight.sum()) - m["arc_weight_sum"]) <= 1e-4 * max(1.0, abs(m["arc_weight_sum"]))
    merged = egs.merge_chain_examples([eg for _, eg in got if eg["outputs"][0]["supervision"].frames_per_sequence == meta["merge_len"]])
    sup = merged["outputs"][0]["supervision"]
    y = synth.random_nnet_output(sup.num_sequences, sup.frames_per_sequence, sup.label_dim, seed=meta["y_seed"])
    assert abs(oracle.num_forward_backward(sup, y)["logprob_weighted"] - meta["merged_num_logprob"]) <= 1e-4 * abs(meta["merged_num_logprob"])


def _same_example(a, b):
    assert [i["name"] for i in a["inputs"]] == [i["name"] for i in b["inputs"]]
    for x, y in zip(a["inputs"], b["inputs"]):
        np.testing.assert_array_equal(x["indexes"], y["indexes"])
        np.testing.assert_array_equal(x["features"], y["features"])  # bit for bit, compressed kinds included
    for x, y in zip(a["outputs"], b["outputs"]):
        assert x["name"] == y["name"] and same_fst(x["supervision"], y["supervision"])
        np.testing.assert_array_equal(x["indexes"], y["indexes"])
        np.testing.assert_array_equal(x["deriv_weights"], y["deriv_weights"])
    assert len(a["outputs"]) == len(b["outputs"])


@pytest.mark.parametrize("kind,dw,e2e", [("FM", "DW2", False), ("DM", "DW", False), ("CM", "DW2", True), ("CM2", "DW", False),
                                         ("CM3", None, False)])
def test_native_reader_equals_the_numpy_reader(tmp_path, kind, dw, e2e):
    """``tc_example_read`` (csrc/egs_reader.cpp; the reference reads and merges natively through Kaldi,
    src/my_lib_example_rand.cpp:35-177) against this package's numpy statement of the same formats: single examples
    as stored, and minibatches merged -- every matrix kind, both deriv-weight encodings, the <End2End> flag, examples
    of more than one sequence, explicit (127-marked) index elements."""
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    keyed = [("utt%d" % i, make_example(fst, 6, seed=40 + i, n_seq=1 + i % 2)) for i in range(5)]
    ark, scp = str(tmp_path / "egs.ark"), str(tmp_path / "egs.scp")
    kw.write_ark(ark, keyed, scp_path=scp, matrix_kind=kind, dw=dw, e2e_flag=e2e)
    where = [(p, off) for _key, p, off in egs.read_scp(scp)]
    for p, off in where:
        _same_example(egs.read_merged_native([(p, off)], merge_single=False), egs.read_scp_entry(p, off))
    for pick in ([0], [0, 2, 4], [1, 3], [4, 3, 2, 1, 0]):
        entries = [where[i] for i in pick]
        want = egs.merge_chain_examples([egs.read_scp_entry(p, off) for p, off in entries])
        _same_example(egs.read_merged_native(entries), want)


def test_native_reader_refuses_what_the_numpy_reader_refuses(tmp_path):
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    good = b"\0B" + kw.chain_example(make_example(fst, 5, seed=3))
    for name, blob in (("truncated", good[:len(good) // 2]), ("text", b"<Nnet3ChainEg> "), ("token", good.replace(b"<NnetIo>", b"<NnetIO>"))):
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        with pytest.raises(egs.EgsFormatError):
            egs.read_merged_native([(path, 0)])
    with pytest.raises(OSError):
        egs.read_merged_native([(str(tmp_path / "absent"), 0)])
    # examples that do not merge: different frames per sequence
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    open(a, "wb").write(good)
    open(b, "wb").write(b"\0B" + kw.chain_example(make_example(fst, 7, seed=4)))
    with pytest.raises(egs.EgsFormatError):
        egs.read_merged_native([(a, 0), (b, 0)])


def test_native_reader_time_bound(tmp_path):
    """A minibatch of 64 one-sequence examples of 150 frames (40-dim input windows, a supervision of a few hundred
    states each) is read, parsed and merged in a few ms -- the numpy reader needs tens -- so that RandExample's
    look-ahead keeps ahead of a 1 ms training step (scripts/time_chain_loss_egs.py on the GPU box)."""
    import time

    fst = synth.random_den_fst(400, 6, 200, seed=2)
    keyed = [("utt%03d" % i, make_example(fst, 150, seed=60 + i, feat_dim=40, ivec_dim=10)) for i in range(64)]
    ark, scp = str(tmp_path / "egs.ark"), str(tmp_path / "egs.scp")
    kw.write_ark(ark, keyed, scp_path=scp)
    where = [(p, off) for _key, p, off in egs.read_scp(scp)]
    egs.read_merged_native(where)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        got = egs.read_merged_native(where)
        best = min(best, time.perf_counter() - t0)
    assert got["outputs"][0]["supervision"].num_sequences == 64 and got["inputs"][0]["features"].shape == (64 * 458, 40)
    assert best <= 0.015, "native read + merge of 64 x 150 frames took %.1f ms" % (best * 1e3)


def test_native_reader_allocates_for_the_data_that_is_there(tmp_path):
    """A size field of a damaged or crafted example must not drive the allocation (a feature matrix header
    may claim 2^31 floats, an FST 2^28 states): the reader's buffers grow with the bytes it really reads, so such a file
    is refused after a few megabytes -- measured here as the process's peak memory."""
    import resource
    import struct
    fst = synth.random_den_fst(40, 4, 24, seed=1)
    good = b"\0B" + kw.chain_example(make_example(fst, 5, seed=3))
    at = good.index(b"FM ") + 3  # [K] WriteBasicType: size byte 4 + int32 rows, size byte 4 + int32 cols
    assert good[at] == 4 and good[at + 5] == 4
    huge = good[:at] + b"\x04" + struct.pack("<i", 1 << 16) + b"\x04" + struct.pack("<i", 1 << 15) + good[at + 10:]
    path = str(tmp_path / "huge")
    open(path, "wb").write(huge)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(egs.EgsFormatError):
        egs.read_merged_native([(path, 0)])
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024.0
    assert grown_mb < 256, "the reader grew by %.0f MB for a %d-byte file" % (grown_mb, len(huge))
