"""Handles of the chain loss: mirrors ``torchain/io.py`` of the reference for the hot path.

``DenominatorGraph`` (reference ``io.py:51-57``) and ``Supervision`` (``io.py:20-31``) keep the
attributes the reference exposes (``.ptr``, ``.n_pdf``, ``.n_batch``, ``.n_frame``, ``.shape``) but
wrap the C-ABI handles of libtorchain_hip.so instead of heap Kaldi objects.  The Kaldi egs readers
of the reference (``Example``, ``RandExample``, ``open_example``, ``io.py:60-175``) are out of
scope: supervisions are built from FST arrays.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib


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

    def prepare(self, device=None):
        """Uploads the immutable tables to ``device`` now instead of on the first loss call."""
        dev = torch.cuda.current_device() if device is None else torch.device(device).index
        check(lib.tc_den_graph_prepare(self.ptr, int(dev)), "tc_den_graph_prepare")
        return self

    def stats(self):
        out = np.zeros(9, np.int64)
        check(lib.tc_den_graph_stats(self.ptr, _p(out)), "tc_den_graph_stats")
        keys = ("fwd_slots", "bwd_slots", "lds_bytes", "threads", "fwd_rows", "bwd_rows", "fwd_conflict_x1000",
                "bwd_conflict_x1000", "tied")
        return dict(zip(keys, (int(x) for x in out)))


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
