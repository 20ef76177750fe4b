"""Handles of the chain loss: mirrors ``torchain/io.py`` of the reference for the hot path.

``DenominatorGraph`` (reference ``io.py:51-57``) and ``Supervision`` (``io.py:20-31``) keep the
attributes the reference exposes (``.ptr``, ``.n_pdf``, ``.n_batch``, ``.n_frame``, ``.shape``) but
wrap the C-ABI handles of libtorchain_hip.so instead of heap Kaldi objects.  The Kaldi egs readers
of the reference (``Example``, ``RandExample``, ``open_example``, ``print_key_length``, ``io.py:60-175``) are
here too, on top of ``torchain_amd.egs`` (a Kaldi-free parser of the ``<Nnet3ChainEg>`` formats).
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor
from contextlib import contextmanager

import numpy as np
import torch

from . import egs as _egs
from ._lib import check, lib


def _tuning_cache_path():
    return os.environ.get("TORCHAIN_TUNING_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "torchain_amd", "tuning.json")


def set_kaldi_device(device_id=0):
    """Reference ``io.py:15-17`` re-points Kaldi's process-wide CuDevice singleton at a torch
    device.  The HIP path is stateless (device and stream are explicit arguments of every call), so
    this only selects the current torch device, which is what callers rely on afterwards."""
    if torch.cuda.is_available():
        torch.cuda.set_device(device_id)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class DenominatorGraph:
    """``DenominatorGraph(rspec, n_pdf)`` as in the reference (``io.py:51-54`` ->
    ``my_lib_denominator_graph_new``, ``src/my_lib_example.cpp:129-134``): ``rspec`` is the path of an
    OpenFst ``den.fst``.  ``rspec`` may also be an in-memory FST: any object with the fields
    ``num_states, src, dst, ilabel, weight, final, start`` (``torchain_amd.synth.DenFst``)."""

    def __init__(self, rspec, n_pdf):
        self.rspec = rspec
        self.n_pdf = int(n_pdf)
        handle = C.c_void_p()
        if isinstance(rspec, (str, bytes)):
            path = rspec.encode() if isinstance(rspec, str) else rspec
            check(lib.tc_den_graph_read(C.byref(handle), path, self.n_pdf), "tc_den_graph_read(%r)" % rspec)
        else:
            f = rspec
            src, dst, il = _i32(f.src), _i32(f.dst), _i32(f.ilabel)
            w, fin = _f32(f.weight), _f32(f.final)
            check(lib.tc_den_graph_create(C.byref(handle), int(f.num_states), len(src), _p(src), _p(dst), _p(il),
                                          _p(w), _p(fin), int(f.start), self.n_pdf), "tc_den_graph_create")
        self.ptr = handle

    def __del__(self, _free=lib.tc_den_graph_free):
        ptr = getattr(self, "ptr", None)
        if ptr:
            _free(ptr)
            self.ptr = None

    @property
    def num_states(self):
        return lib.tc_den_graph_num_states(self.ptr)

    @property
    def num_arcs(self):
        return lib.tc_den_graph_num_arcs(self.ptr)

    def initial_probs(self):
        out = np.zeros(self.num_states, np.float32)
        check(lib.tc_den_graph_initial_probs(self.ptr, _p(out)), "tc_den_graph_initial_probs")
        return out

    def prepare(self, device=None, variant=None):
        """Uploads the immutable tables to ``device`` now instead of on the first loss call, and settles which of the
        graph's two kernels batches above one sequence per two CUs run there (``tc_den_graph_tuning``): ``variant``
        (0 fused / 1 two-sequence) if given -- ``parallel.sync_den_graph_variant`` passes rank 0's --, else the choice
        an earlier run measured for this graph on this kind of device (a JSON cache: ``$TORCHAIN_TUNING_CACHE`` or
        ``~/.cache/torchain_amd/tuning.json``, keyed by ``tc_den_graph_hash`` and the device name), else the library
        times the two kernels once and the result goes into that cache.  So the choice -- and with it the last bits of
        the results -- is the same from run to run and from rank to rank.  Since round 5 the cache is the LIBRARY's
        (``csrc/tuning_cache.cpp``, inside ``tc_den_graph_prepare``): a caller of the C ABI gets the same without this
        class."""
        dev = torch.cuda.current_device() if device is None else torch.device(device).index
        dev = int(dev or 0)
        done = self.__dict__.setdefault("_prepared", {})
        if dev in done and variant is None:
            return self
        if variant is not None:
            check(lib.tc_den_graph_set_variant(self.ptr, dev, int(variant)), "tc_den_graph_set_variant")
        check(lib.tc_den_graph_prepare(self.ptr, dev), "tc_den_graph_prepare")  # (looks the graph up in the cache, or times)
        done[dev] = True
        return self

    def stats(self):
        out = np.zeros(9, np.int64)
        check(lib.tc_den_graph_stats(self.ptr, _p(out)), "tc_den_graph_stats")
        keys = ("fwd_slots", "bwd_slots", "lds_bytes", "threads", "fwd_rows", "bwd_rows", "fwd_conflict_x1000",
                "bwd_conflict_x1000", "tied")
        return dict(zip(keys, (int(x) for x in out)))


    def tuning(self, device):
        """Which kernel batches above one sequence per two CUs run on ``device`` (``tc_den_graph_tuning``) and the
        two times the choice was made on."""
        import ctypes as C
        dev = device if isinstance(device, int) else torch.device(device).index
        choice, a, b = C.c_int32(0), C.c_float(0), C.c_float(0)
        check(lib.tc_den_graph_tuning(self.ptr, int(dev or 0), C.byref(choice), C.byref(a), C.byref(b)),
              "tc_den_graph_tuning")
        return {"two_sequence_kernel": int(choice.value), "fused_ms": float(a.value), "two_sequence_ms": float(b.value)}

    def debug_walk(self, direction, gather, pdf_factor):
        """Diagnostic (host only): replays one arc walk from the built schedules, see
        ``tc_den_graph_debug_walk``.  Used by the CPU tests of the schedule builder."""
        gather = np.ascontiguousarray(gather, np.float32)
        pdf_factor = np.ascontiguousarray(pdf_factor, np.float32)
        out = np.zeros(lib.tc_den_graph_num_states(self.ptr), np.float32)
        check(lib.tc_den_graph_debug_walk(self.ptr, int(direction), _p(gather), _p(pdf_factor), _p(out)),
              "tc_den_graph_debug_walk")
        return out


class Supervision:
    """Reference ``io.py:20-31``: wraps a supervision handle and exposes ``n_pdf``, ``n_batch``,
    ``n_frame``, ``shape``.  A null handle raises ``ValueError`` exactly like the reference (its
    iterators skip such batches, ``io.py:105-110``)."""

    def __init__(self, ptr):
        self.ptr = ptr if isinstance(ptr, C.c_void_p) else C.c_void_p(ptr)
        if not self.ptr:
            raise ValueError("null supervision ptr")
        self.n_pdf = lib.tc_supervision_num_pdf(self.ptr)
        self.n_batch = lib.tc_supervision_num_sequence(self.ptr)
        self.n_frame = lib.tc_supervision_num_frame(self.ptr)
        self.shape = (self.n_batch, self.n_frame, self.n_pdf)

    @classmethod
    def from_fst(cls, weight, num_sequences, frames_per_sequence, label_dim, arc_begin, ilabel, arc_weight,
                 nextstate, final):
        """Builds the handle from the five fields of [K] chain::Supervision the path uses
        (``src/my_lib_example.cpp:82-95``); the FST is the merged acceptor in CSR form."""
        ab, il, nx = _i32(arc_begin), _i32(ilabel), _i32(nextstate)
        aw, fin = _f32(arc_weight), _f32(final)
        handle = C.c_void_p()
        check(lib.tc_supervision_create(C.byref(handle), float(weight), int(num_sequences), int(frames_per_sequence),
                                        int(label_dim), len(fin), _p(ab), _p(il), _p(aw), _p(nx), _p(fin)),
              "tc_supervision_create")
        return cls(handle)

    @classmethod
    def from_synth(cls, sup):
        """From a ``torchain_amd.synth.SupFst``."""
        return cls.from_fst(sup.weight, sup.num_sequences, sup.frames_per_sequence, sup.label_dim, sup.arc_begin,
                            sup.ilabel, sup.arc_weight, sup.nextstate, sup.final)

    @property
    def weight(self):
        return lib.tc_supervision_weight(self.ptr)

    def __del__(self, _free=lib.tc_supervision_free):
        ptr = getattr(self, "ptr", None)
        if ptr:
            _free(ptr)
            self.ptr = None


# ---- Kaldi chain-egs readers (reference io.py:60-175) ------------------------------------------------------
class Example:
    """``Example(rspec)``: sequential reader of chain egs (reference ``io.py:60-113`` over Kaldi's
    SequentialNnetChainExampleReader, ``src/my_lib_example.cpp:35-127``).  ``rspec`` is a Kaldi rspecifier
    (``ark:file``, ``ark,bg:file``, ``scp:file``, ``ark:command |``).  Same surface as the reference: ``next()``,
    ``load_feats``, ``supervision``, ``indexes``, ``inputs``, ``value()``, iteration yielding
    ``((inp (B, F, T_in), aux), Supervision)``.

    One deliberate difference: the reference's ``next()`` advances a reader that already stands on the first
    example, so its iteration silently drops the first example of every archive (and ends on a null supervision);
    here every example is delivered.  ``deriv_weights`` -- which the reference drops (``README.md:41``) -- is
    available as a property."""

    def __init__(self, rspec):
        self.rspec = rspec
        self._it = _egs.iter_rspecifier(rspec)
        self._cur = None

    def next(self):
        try:
            _key, eg = next(self._it)
        except StopIteration:
            self._cur = None
            return False
        self._cur = eg
        return True

    def _need(self):
        if self._cur is None:
            raise ValueError("null supervision ptr")  # what the reference raises past the end (io.py:23-24)
        return self._cur

    def load_feats(self, inp=None, aux=None):
        """Reference ``my_lib_example_feats``: number of inputs of the current example (tensors are returned by
        ``inputs``; the in-place TH resize of the reference has no counterpart)."""
        return len(self._need()["inputs"])

    @property
    def supervision(self):
        cur = self._need()
        if "_handle" not in cur:  # (RandExample's look-ahead has usually built it already)
            cur["_handle"] = Supervision.from_synth(cur["outputs"][0]["supervision"])
        return cur["_handle"]

    @property
    def indexes(self):
        """(B, T) LongTensor of the output frames' ``t`` (reference ``my_lib_example_reader_indexes``)."""
        out = self._need()["outputs"][0]
        sup = out["supervision"]
        t = out["indexes"][:, 1].reshape(sup.frames_per_sequence, sup.num_sequences)
        return torch.from_numpy(np.ascontiguousarray(t.T).astype(np.int64))

    @property
    def deriv_weights(self):
        """(T*B,) float tensor, frame-major like the nnet output rows ([K] NnetChainSupervision::deriv_weights)."""
        return torch.from_numpy(self._need()["outputs"][0]["deriv_weights"].astype(np.float32))

    @property
    def inputs(self):
        ins = self._need()["inputs"]
        if len(ins) == 1:
            return torch.from_numpy(ins[0]["features"]), None
        if len(ins) == 2:
            return torch.from_numpy(ins[0]["features"]), torch.from_numpy(ins[1]["features"])
        raise ValueError("unsupported number of inputs (up to 2): %d" % len(ins))

    def value(self):
        supervision = self.supervision
        n_batch, n_out_frame, n_pdf = supervision.shape
        inp, aux = self.inputs
        if inp is not None:
            inp = inp.view(n_batch, -1, inp.shape[1]).transpose(1, 2)
        return (inp, aux), supervision

    def __iter__(self):
        while self.next():
            try:
                yield self.value()
            except ValueError:
                continue


def feats(egs):
    """Reference ``io.py:34-48`` (broken there: it refers to ``self``): (input, aux-or-None) of the current example."""
    if isinstance(egs, Example):
        return egs.inputs
    raise ValueError("unknown reader type")


@contextmanager
def open_example(cmd):
    """Reference ``io.py:116-131``: runs ``cmd`` (e.g. ``nnet3-chain-copy-egs ... ark:-``) and reads the egs it writes
    to stdout.  The reference goes through a FIFO and Kaldi's ``ark,bg:`` reader; a pipe does the same job."""
    set_kaldi_device()
    example = Example("ark:" + cmd + " |")
    try:
        yield example
    finally:
        del example


def print_key_length(scp_path, len_file="/dev/stdout"):
    """Reference ``io.py:134-135`` / ``my_lib_example_rand.cpp:168-177``: ``key frames_per_sequence`` per example."""
    with open(len_file, "w") as f:
        for key, eg in _egs.iter_rspecifier(scp_path):
            f.write("%s %d\n" % (key, eg["outputs"][0]["supervision"].frames_per_sequence))


class RandExample(Example):
    """``RandExample(scp_path, seed, batchsize, len_file="")``: random-access reader that groups examples of equal
    ``frames_per_sequence`` into minibatches, shuffles inside and across groups and merges each minibatch
    ([K] MergeChainExamples) -- reference ``io.py:138-175`` over ``src/my_lib_example_rand.cpp:35-177``.  The
    length file (``scp_path + ".len"`` unless given) holds ``key length`` pairs; without it the lengths are read
    from the egs.  The shuffle of the default (native) reader is a Fisher-Yates pass over ``std::mt19937`` draws
    (``csrc/rand_reader.cpp``): the reference's generator, but not the draw order of its ``std::shuffle``, whose algorithm
    is not specified -- and not the order round 3's Python reader (``native=False``: NumPy's MT19937 shuffle) gives for
    the same ``seed``: the batch order for a given seed changed when the native reader became the default in round 4."""

    def __init__(self, scp_path, seed, batchsize, len_file="", prefetch=True, rank=0, world=1, native=True, device=None,
                 order="sorted"):
        """``rank`` / ``world``: this process's share of a data-parallel job (every rank forms the same shuffled list of
        minibatches from ``seed`` and takes every ``world``-th; all ranks get the same number).  ``native`` (default): the
        whole reader -- bucketing, shuffle, look-ahead threads, merge, supervision handles -- is the library's
        ``tc_rand_reader_*`` handle, as the reference's is ``my_lib_example_rand_reader_*`` (``src/my_lib.h:8-17``);
        ``device`` is the GPU the batches are for: given explicitly, the look-ahead threads also fill the pinned staging of
        each supervision's upload (they then touch that GPU: ``hipSetDevice`` / pinned allocations on worker threads); left
        ``None``, building the reader touches no GPU -- a script may still fork or re-launch itself -- and the staging starts
        with the first ``next()`` made after the process has initialised CUDA, for the then current device.
        ``native=False`` keeps the Python statement of it below (NumPy's MT19937 shuffle, a thread pool), which needs
        ``rank == 0, world == 1``.
        ``order``: ``"sorted"`` (default; lengths ascending, a Fisher-Yates pass that is the same on every platform) or
        ``"reference"``: the reference's ``shuffle_keys`` restated on the same standard-library calls
        (``src/my_lib_example_rand.cpp:119-141``: ``std::unordered_map`` iteration, ``std::shuffle`` over ``std::mt19937``) --
        the batch lists a reference built against libstdc++ forms for the same ``seed``, epoch for epoch (native reader
        only; ``rank`` / ``world`` sharding applies on top)."""
        assert os.path.exists(scp_path)
        self.scp_path = scp_path
        self.rspec = scp_path
        self.batchsize = int(batchsize)
        self._native = None
        if native:
            handle = C.c_void_p()
            # (True = 3 look-ahead threads: a batch of 64 x 150 frames costs one of them ~1.5 ms, a step 0.7 ms; with more
            # the training thread's own enqueue slows down -- 0.11 ms beside two readers, 0.39 beside four, 0.47 beside
            # eight on the test box -- and the step with it: profiles/r04_egs_steps.txt)
            depth = (3 if prefetch is True else int(prefetch)) if prefetch else 0
            if order not in ("sorted", "reference"):
                raise ValueError("order must be 'sorted' or 'reference'")
            rc = lib.tc_rand_reader_new_ordered(os.fsencode(scp_path), int(seed), int(batchsize), os.fsencode(len_file or ""),
                                                int(rank), int(world), depth, 1 if order == "reference" else 0, C.byref(handle))
            if rc != 0:
                raise _egs.EgsFormatError((lib.tc_rand_reader_last_error() or b"").decode() or "tc_rand_reader_new: %d" % rc)
            self._native = handle
            self._cur = None
            self._have = False
            # the GPU the minibatches are for: the look-ahead threads then also fill the pinned staging of each supervision's
            # upload, which otherwise falls to the training thread's first use of it.  Only an EXPLICIT device is set here:
            # asking torch for the current one would initialise the GPU context in a constructor (see the docstring).
            self._staging_device = None
            if device is not None:
                self._set_staging_device(int(torch.device("cuda", device).index if not isinstance(device, int) else device))
            return
        if rank != 0 or world != 1 or order != "sorted":
            raise ValueError("the Python reader does not shard and has one order: use native=True")
        # The next minibatches are prepared on background threads while the current one is in use (reading, parsing,
        # merging and building the supervision handle are library calls that release the interpreter lock): the
        # training thread finds its batch ready instead of re-opening and re-parsing every scp entry synchronously.
        # One batch costs about 1.6 ms of such work for 64 x 150 frames, a training-side step 0.7 ms: ``prefetch``
        # batches (True = 8) are kept under way, each on its own thread.
        self._depth = (8 if prefetch is True else int(prefetch)) if prefetch else 0
        self._pool = ThreadPoolExecutor(max_workers=self._depth) if self._depth else None
        self._pending = {}  # position -> future
        self._rng = np.random.RandomState(int(seed))
        self._where = {key: (p, off) for key, p, off in _egs.read_scp(scp_path)}
        self._length_to_keys = {}
        lf = len_file or scp_path + ".len"
        if os.path.exists(lf):
            toks = open(lf).read().split()
            for key, length in zip(toks[0::2], toks[1::2]):
                self._length_to_keys.setdefault(int(length), []).append(key)
        else:
            for key, (p, off) in self._where.items():
                eg = _egs.read_scp_entry(p, off)
                self._length_to_keys.setdefault(eg["outputs"][0]["supervision"].frames_per_sequence, []).append(key)
        self._n_data = sum(len(v) for v in self._length_to_keys.values())
        self._shuffle_keys()
        self._pos = -1
        self._cur = None

    def _set_staging_device(self, index):
        check(lib.tc_rand_reader_set_device(self._native, int(index)), "tc_rand_reader_set_device")
        self._staging_device = int(index)

    def _shuffle_keys(self):
        self._key_batch = []
        for length in sorted(self._length_to_keys):
            keys = list(self._length_to_keys[length])
            self._rng.shuffle(keys)
            for i in range(0, len(keys), self.batchsize):
                self._key_batch.append(keys[i:i + self.batchsize])
        self._rng.shuffle(self._key_batch)

    def reset(self):
        if self._native is not None:
            check(lib.tc_rand_reader_reset(self._native), "tc_rand_reader_reset")
            self._cur, self._have = None, False
            return
        self._drop_pending()
        self._pos = -1
        self._cur = None
        self._shuffle_keys()

    def _drop_pending(self):
        for fut in self._pending.values():
            fut.cancel()
        for fut in self._pending.values():
            try:
                fut.result()
            except Exception:  # (a cancelled or failed look-ahead is simply not used)
                pass
        self._pending = {}

    def _load(self, pos):
        # read + parse + merge in the library (tc_example_read), as the reference does in Kaldi, and the supervision
        # handle (tc_supervision_create: per-sequence split, time levels; 1.2 ms of host work for 64 x 150 frames) with
        # it: on the look-ahead thread both run beside the training step, outside the interpreter lock
        merged = _egs.read_merged_native([self._where[k] for k in self._key_batch[pos]])
        try:
            merged["_handle"] = Supervision.from_synth(merged["outputs"][0]["supervision"])
        except Exception:  # (reported when the batch is used: Example.supervision builds it again and raises there)
            pass
        return merged

    def __del__(self, _free=lib.tc_rand_reader_free):
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)
        handle = getattr(self, "_native", None)
        if handle:
            _free(handle)
            self._native = None

    @property
    def n_batch(self):
        if self._native is not None:
            return lib.tc_rand_reader_num_batch(self._native)
        return len(self._key_batch)

    @property
    def n_data(self):
        if self._native is not None:
            return lib.tc_rand_reader_num_data(self._native)
        return self._n_data

    def batch_keys(self, batch):
        """The keys of this rank's minibatch ``batch`` of the current epoch (native reader)."""
        buf = C.create_string_buffer(1 << 16)
        n = lib.tc_rand_reader_batch_keys(self._native, int(batch), buf, len(buf))
        check(min(n, 0), "tc_rand_reader_batch_keys")
        return buf.value.decode().split()

    def _need(self):
        if self._native is not None and self._cur is None and self._have:
            # the minibatch the library holds, as the dict the numpy reader returns; the supervision handle the
            # look-ahead thread built comes with it
            sup = C.c_void_p()
            check(lib.tc_rand_reader_supervision_new(self._native, C.byref(sup)), "tc_rand_reader_supervision_new")
            handle = Supervision(sup)
            eg = C.c_void_p()
            check(lib.tc_rand_reader_take_example(self._native, C.byref(eg)), "tc_rand_reader_take_example")
            self._cur = _egs._wrap_native(eg)
            self._cur["_handle"] = handle
        return super()._need()

    def next(self):
        if self._native is not None:
            # (is_initialized() first: on ROCm is_available() itself initialises the runtime, and a reader built with
            # device=None promises to touch no GPU until the process has)
            if self._staging_device is None and torch.cuda.is_initialized():
                self._set_staging_device(torch.cuda.current_device())  # (the process uses the GPU by now: no new context)
            rc = lib.tc_rand_reader_next(self._native)
            self._cur = None
            self._have = rc == 1
            if rc < 0:
                raise _egs.EgsFormatError((lib.tc_rand_reader_last_error() or b"").decode() or "tc_rand_reader_next: %d" % rc)
            return rc == 1
        self._pos += 1
        if self._pos >= len(self._key_batch):
            self._cur = None
            return False
        fut = self._pending.pop(self._pos, None)
        self._cur = fut.result() if fut is not None else self._load(self._pos)
        if self._pool is not None:
            for pos in range(self._pos + 1, min(self._pos + 1 + self._depth, len(self._key_batch))):
                if pos not in self._pending:
                    self._pending[pos] = self._pool.submit(self._load, pos)
        return True
