"""ctypes binding of oracle/libchain_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (torchain_amd/) never does.  See the header of chain_oracle.c for what is restated
and why parity is unpinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libchain_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "chain_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libchain_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_den_graph_new.restype = C.c_void_p
        L.oracle_den_graph_new.argtypes = [C.c_int32, C.c_int64, _i32p, _i32p, _i32p, _f32p, _f32p, C.c_int32, C.c_int32]
        L.oracle_den_graph_free.argtypes = [C.c_void_p]
        L.oracle_den_graph_initial_probs.argtypes = [C.c_void_p, _f32p]
        L.oracle_den_graph_transitions.argtypes = [C.c_void_p, _f32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p]
        L.oracle_den_forward_backward.restype = C.c_int
        L.oracle_den_forward_backward.argtypes = [
            C.c_void_p, C.c_float, C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p,
            C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_num_forward_backward.restype = C.c_int
        L.oracle_num_forward_backward.argtypes = [
            C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f32p, _i32p, _f32p, _f32p,
            C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_float)]
        L.oracle_compute_chain_objf_and_deriv.restype = C.c_int
        L.oracle_compute_chain_objf_and_deriv.argtypes = [
            C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
            _i32p, _i32p, _f32p, _i32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int64, _f32p, C.c_void_p,
            C.c_int64, C.c_void_p, C.c_int64]
        L.oracle_den_forward_backward_blocks.restype = C.c_int
        L.oracle_den_forward_backward_blocks.argtypes = [
            C.c_void_p, C.c_float, C.c_int32, C.c_int32, _f32p, C.c_int64, C.c_float, C.c_void_p, C.c_int64,
            C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.oracle_fst_state_times.restype = C.c_int32
        L.oracle_fst_state_times.argtypes = [C.c_int32, _i32p, _i32p, _i32p, _f32p, _i32p]
        L.oracle_set_exp_clamp.argtypes = [C.c_int]
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class DenGraph:
    """[K] DenominatorGraph built from flat FST arrays (torchain_amd.synth.DenFst)."""

    def __init__(self, fst):
        self.fst = fst
        self.ptr = lib().oracle_den_graph_new(
            fst.num_states, len(fst.src), _c(fst.src, np.int32), _c(fst.dst, np.int32), _c(fst.ilabel, np.int32),
            _c(fst.weight, np.float32), _c(fst.final, np.float32), fst.start, fst.num_pdfs)
        if not self.ptr:
            raise ValueError("oracle_den_graph_new rejected the FST")
        self.num_states, self.num_pdfs, self.num_arcs = fst.num_states, fst.num_pdfs, len(fst.src)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                lib().oracle_den_graph_free(self.ptr)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self.ptr = None

    def initial_probs(self):
        out = np.zeros(self.num_states, np.float32)
        lib().oracle_den_graph_initial_probs(self.ptr, out)
        return out

    def transitions(self):
        A, H = self.num_arcs, self.num_states
        prob, pdf, st = np.zeros(2 * A, np.float32), np.zeros(2 * A, np.int32), np.zeros(2 * A, np.int32)
        idx = [np.zeros(H, np.int32) for _ in range(4)]
        lib().oracle_den_graph_transitions(self.ptr, prob, pdf, st, *idx)
        return prob, pdf, st, idx


def den_forward_backward(graph, y, num_sequences, leaky=1e-5, deriv_weight=1.0, want_deriv=True):
    """[K] DenominatorComputation::Forward (+ Backward(deriv_weight, &deriv) into a zero matrix).
    Returns dict(logprob, deriv, ok, alpha_beta, gamma_sum)."""
    y = _c(y, np.float32)
    rows, cols = y.shape
    deriv = np.zeros_like(y) if want_deriv else None
    lp, ok, ab, gs = C.c_float(0), C.c_int32(1), C.c_float(0), C.c_float(0)
    rc = lib().oracle_den_forward_backward(
        graph.ptr, leaky, num_sequences, y, rows, cols, cols, deriv_weight,
        deriv.ctypes.data if want_deriv else None, cols, C.byref(lp), C.byref(ok), C.byref(ab), C.byref(gs))
    if rc:
        raise ValueError("oracle_den_forward_backward rc=%d" % rc)
    return dict(logprob=lp.value, deriv=deriv, ok=bool(ok.value), alpha_beta=ab.value, gamma_sum=gs.value)


def num_forward_backward(sup, y, want_deriv=True):
    """[K] NumeratorComputation::Forward/Backward on the merged supervision FST."""
    y = _c(y, np.float32)
    rows, cols = y.shape
    deriv = np.zeros_like(y) if want_deriv else None
    lp = C.c_float(0)
    rc = lib().oracle_num_forward_backward(
        sup.weight, sup.num_sequences, sup.frames_per_sequence, sup.label_dim, sup.num_states,
        _c(sup.arc_begin, np.int32), _c(sup.ilabel, np.int32), _c(sup.arc_weight, np.float32),
        _c(sup.nextstate, np.int32), _c(sup.final, np.float32), y, rows, cols, cols,
        deriv.ctypes.data if want_deriv else None, cols, C.byref(lp))
    if rc:
        raise ValueError("oracle_num_forward_backward rc=%d" % rc)
    return dict(logprob_weighted=lp.value, deriv=deriv)


def compute_chain_objf_and_deriv(graph, sup, y, l2_regularize=0.0, leaky_hmm_coefficient=1e-5, xent_regularize=0.0,
                                 want_deriv=True, want_xent=False):
    """[K] ComputeChainObjfAndDeriv; results = [objf, l2_term, weight] (my_lib_chain.cpp:126,130)."""
    y = _c(y, np.float32)
    rows, cols = y.shape
    deriv = np.empty_like(y) if want_deriv else None
    xent = np.empty_like(y) if want_xent else None
    res = np.zeros(3, np.float32)
    rc = lib().oracle_compute_chain_objf_and_deriv(
        graph.ptr, l2_regularize, leaky_hmm_coefficient, xent_regularize, sup.weight, sup.num_sequences,
        sup.frames_per_sequence, sup.label_dim, sup.num_states, _c(sup.arc_begin, np.int32),
        _c(sup.ilabel, np.int32), _c(sup.arc_weight, np.float32), _c(sup.nextstate, np.int32),
        _c(sup.final, np.float32), y, rows, cols, cols, res, deriv.ctypes.data if want_deriv else None, cols,
        xent.ctypes.data if want_xent else None, cols)
    if rc:
        raise ValueError("oracle_compute_chain_objf_and_deriv rc=%d" % rc)
    return dict(objf=float(res[0]), l2_term=float(res[1]), weight=float(res[2]), results=res, deriv=deriv,
                xent_deriv=xent)


def den_forward_backward_blocks(graph, y, num_sequences, frames, leaky, block=1, threads=1, deriv_weight=1.0,
                                want_deriv=True):
    y = _c(y, np.float32)
    deriv = np.zeros_like(y) if want_deriv else None
    tot = C.c_double(0)
    rc = lib().oracle_den_forward_backward_blocks(
        graph.ptr, leaky, num_sequences, frames, y, y.shape[1], deriv_weight,
        deriv.ctypes.data if want_deriv else None, y.shape[1], block, threads, C.byref(tot))
    if rc:
        raise ValueError("rc=%d" % rc)
    return tot.value, deriv


def fst_state_times(sup):
    st = np.zeros(sup.num_states, np.int32)
    total = lib().oracle_fst_state_times(sup.num_states, _c(sup.arc_begin, np.int32), _c(sup.ilabel, np.int32),
                                         _c(sup.nextstate, np.int32), _c(sup.final, np.float32), st)
    return total, st
