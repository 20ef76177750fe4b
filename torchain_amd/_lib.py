"""ctypes binding of libtorchain_hip.so, the C-ABI drop-in boundary (include/torchain_hip.h).

The reference binds its C functions through cffi (``torchain/functions.py:5-6``,
``build.py:19-31``); here the same role is played by ctypes on a hipcc-built shared library that
takes raw device pointers.  There is no CPU fallback: if the library is missing the import of this
module raises, and every hot call requires CUDA/ROCm tensors.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtorchain_hip.so")

TC_OK = 0
ERRORS = {
    -1: "TC_ERR_INVALID_ARGUMENT", -2: "TC_ERR_BAD_FST", -3: "TC_ERR_UNSUPPORTED", -4: "TC_ERR_WORKSPACE",
    -5: "TC_ERR_HIP", -6: "TC_ERR_IO", -7: "TC_ERR_NOT_SEPARABLE",
}


class TorchainHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.tc_strerror(code).decode() if _lib is not None else "?"
        extra = " (hipError %d)" % lib.tc_last_hip_error() if code == -5 else ""
        super().__init__("%s failed: %s [%s]%s" % (where, msg, ERRORS.get(code, code), extra))


_lib = None


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "torchain_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C torchain_amd/csrc`; there is no CPU fallback for the chain loss." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.tc_strerror.restype = C.c_char_p
    L.tc_strerror.argtypes = [C.c_int]
    L.tc_version.restype = C.c_int
    L.tc_last_hip_error.restype = C.c_int
    L.tc_den_graph_create.restype = C.c_int
    L.tc_den_graph_create.argtypes = [C.POINTER(vp), i32, i64, vp, vp, vp, vp, vp, i32, i32]
    L.tc_den_graph_read.restype = C.c_int
    L.tc_den_graph_read.argtypes = [C.POINTER(vp), C.c_char_p, i32]
    L.tc_den_graph_free.restype = None
    L.tc_den_graph_free.argtypes = [vp]
    L.tc_den_graph_num_states.restype = i32
    L.tc_den_graph_num_states.argtypes = [vp]
    L.tc_den_graph_num_arcs.restype = i64
    L.tc_den_graph_num_arcs.argtypes = [vp]
    L.tc_den_graph_num_pdfs.restype = i32
    L.tc_den_graph_num_pdfs.argtypes = [vp]
    L.tc_den_graph_initial_probs.restype = C.c_int
    L.tc_den_graph_initial_probs.argtypes = [vp, vp]
    L.tc_supervision_append.restype = C.c_int
    L.tc_supervision_append.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.tc_self_test.restype = C.c_int
    L.tc_self_test.argtypes = [C.c_int, vp, vp]
    L.tc_den_graph_prepare.restype = C.c_int
    L.tc_den_graph_prepare.argtypes = [vp, C.c_int]
    L.tc_den_graph_stats.restype = C.c_int
    L.tc_den_graph_stats.argtypes = [vp, vp]
    L.tc_example_read.restype = C.c_int
    L.tc_example_read.argtypes = [vp, vp, C.c_int32, C.c_int, vp]
    L.tc_archive_open.restype = C.c_int
    L.tc_archive_open.argtypes = [C.c_char_p, vp]
    L.tc_archive_next.restype = C.c_int
    L.tc_archive_next.argtypes = [vp, vp, C.c_int32, vp]
    L.tc_archive_close.restype = C.c_int
    L.tc_archive_close.argtypes = [vp]
    L.tc_example_free.restype = None
    L.tc_example_free.argtypes = [vp]
    L.tc_example_last_error.restype = C.c_char_p
    L.tc_example_last_error.argtypes = []
    L.tc_example_counts.restype = C.c_int
    L.tc_example_counts.argtypes = [vp, vp]
    L.tc_example_input.restype = C.c_int
    L.tc_example_input.argtypes = [vp, C.c_int32] + [vp] * 6
    L.tc_example_output.restype = C.c_int
    L.tc_example_output.argtypes = [vp, C.c_int32] + [vp] * 11
    L.tc_den_graph_tuning.restype = C.c_int
    L.tc_den_graph_tuning.argtypes = [vp, C.c_int, vp, vp, vp]
    L.tc_den_graph_set_variant.restype = C.c_int
    L.tc_den_graph_set_variant.argtypes = [vp, C.c_int, C.c_int32]
    L.tc_den_graph_hash.restype = C.c_uint64
    L.tc_den_graph_hash.argtypes = [vp]
    L.tc_tuning_cache_get.restype = C.c_int
    L.tc_tuning_cache_get.argtypes = [C.c_uint64, C.c_char_p, C.POINTER(C.c_int32)]
    L.tc_tuning_cache_put.restype = C.c_int
    L.tc_tuning_cache_put.argtypes = [C.c_uint64, C.c_char_p, C.c_int32, C.c_float, C.c_float]
    L.tc_debug_set.restype = C.c_int
    L.tc_debug_set.argtypes = [C.c_char_p, C.c_int]
    L.tc_debug_counter.restype = C.c_int64
    L.tc_debug_counter.argtypes = [C.c_char_p]
    L.tc_den_graph_debug_walk.restype = C.c_int
    L.tc_den_graph_debug_walk.argtypes = [vp, C.c_int, vp, vp, vp]
    L.tc_to2d.restype = C.c_int
    L.tc_to2d.argtypes = [vp, i32, i32, i32, vp, i64, C.c_int, vp]
    L.tc_from2d.restype = C.c_int
    L.tc_from2d.argtypes = [vp, i64, i32, i32, i32, f32, vp, C.c_int, vp]
    L.tc_supervision_create.restype = C.c_int
    L.tc_supervision_create.argtypes = [C.POINTER(vp), f32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.tc_rand_reader_new.restype = C.c_int
    L.tc_rand_reader_new.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, vp]
    L.tc_rand_reader_new_ordered.restype = C.c_int
    L.tc_rand_reader_new_ordered.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    for name in ("tc_rand_reader_reset", "tc_rand_reader_num_batch", "tc_rand_reader_num_data", "tc_rand_reader_next"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    for name in ("tc_rand_reader_example", "tc_rand_reader_supervision_new", "tc_rand_reader_take_example"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp, vp]
    L.tc_rand_reader_set_device.restype = C.c_int
    L.tc_rand_reader_set_device.argtypes = [vp, C.c_int]
    L.tc_supervision_stage.restype = C.c_int
    L.tc_supervision_stage.argtypes = [vp, C.c_int]
    L.tc_rand_reader_batch_keys.restype = C.c_int
    L.tc_rand_reader_batch_keys.argtypes = [vp, C.c_int32, C.c_char_p, C.c_int32]
    L.tc_rand_reader_free.restype = None
    L.tc_rand_reader_free.argtypes = [vp]
    L.tc_rand_reader_last_error.restype = C.c_char_p
    L.tc_rand_reader_last_error.argtypes = []
    L.tc_chain_step_workspace_bytes.restype = C.c_int64
    L.tc_chain_step_workspace_bytes.argtypes = [vp, C.c_int32, C.c_int32, C.c_int, C.c_int]
    L.tc_chain_step.restype = C.c_int
    L.tc_chain_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp,
                                vp, vp, C.c_int64, C.c_int, vp]
    L.tc_supervision_free.restype = None
    L.tc_supervision_free.argtypes = [vp]
    for name in ("tc_supervision_num_pdf", "tc_supervision_num_sequence", "tc_supervision_num_frame"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp]
    L.tc_supervision_weight.restype = f32
    L.tc_supervision_weight.argtypes = [vp]
    L.tc_supervision_prepare.restype = C.c_int
    L.tc_supervision_prepare.argtypes = [vp, C.c_int, vp]
    L.tc_chain_workspace_bytes.restype = i64
    L.tc_chain_workspace_bytes.argtypes = [vp, i32, i32]
    L.tc_chain_objf_and_deriv.restype = C.c_int
    L.tc_chain_objf_and_deriv.argtypes = [vp, vp, vp, i64, i32, i64, vp, vp, i64, vp, i64, f32, f32, f32, vp, i64,
                                          C.c_int, vp]
    L.tc_chain_objf_and_grad.restype = C.c_int
    L.tc_chain_objf_and_grad.argtypes = L.tc_chain_objf_and_deriv.argtypes
    L.tc_den_forward_backward.restype = C.c_int
    L.tc_den_forward_backward.argtypes = [vp, i32, vp, i64, i32, i64, f32, f32, f32, C.c_int, vp, i64, vp, vp, vp,
                                          i64, C.c_int, vp]
    L.tc_xent_objf.restype = C.c_int
    L.tc_xent_objf.argtypes = [vp, i64, i32, i64, vp, i64, vp, vp, i64, C.c_int, vp]
    L.tc_num_forward_backward.restype = C.c_int
    L.tc_num_forward_backward.argtypes = [vp, vp, i64, i32, i64, vp, i64, vp, vp, i64, C.c_int, vp]
    return L


lib = _load()
_lib = lib


def check(code, where):
    if code != TC_OK:
        raise TorchainHipError(code, where)
