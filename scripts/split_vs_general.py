"""Development aid (GPU box): nearly chain-structured graphs -- state splitting + tied kernel vs the general kernel on the
unsplit graph (tc_debug_set("no_split")), fwd+bwd ms per 256 x 150 batch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402


def time_graph(fst, P, S=256, T=150, flags=()):
    for f in ("no_split", "old_general"):
        lib.tc_debug_set(f.encode(), 1 if f in flags else 0)
    lib.tc_debug_set(b"no_tune", 1)
    dev = torch.device("cuda", 0)
    g = io.DenominatorGraph(fst, P).prepare(dev)
    st = g.stats()
    y = torch.randn(S * T, P, device=dev)
    d = torch.empty_like(y)
    nb = lib.tc_chain_workspace_bytes(g.ptr, S, T)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()

    def run():
        check(lib.tc_den_forward_backward(g.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), 0.1, -1.0, 0.0, 0,
                                          C.c_void_p(d.data_ptr()), d.stride(0), None, None, C.c_void_p(ws.data_ptr()), nb, 0,
                                          C.c_void_p(stream.cuda_stream)), "den")
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 10, st


cases = [("R3", synth.config_den_fst("R3"), 2928)]
for H, frac in ((3000, 0.6), (3000, 0.9), (6000, 0.6), (6000, 0.9), (8000, 0.3)):
    cases.append(("nearly H=%d frac=%.1f" % (H, frac), synth.nearly_tied_den_fst(H, 8, 2000, seed=42, fraction=frac), 2000))
for name, fst, P in cases:
    a, sa = time_graph(fst, P)
    b, sb = time_graph(fst, P, flags=("no_split",))
    c, sc = time_graph(fst, P, flags=("no_split", "old_general"))
    print("%-24s states %5d arcs %6d | split: kernel %d lds %6d %6.3f ms | general (round 5): kernel %d %6.3f ms | general (round 1): %6.3f ms" % (
        name, fst.num_states, len(fst.src), sa["tied"], sa["lds_bytes"], a, sb["tied"], b, c))
