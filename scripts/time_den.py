"""Times tc_den_forward_backward variants on one GPU (development aid, not part of the contract)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402

for key in os.environ.get("TC_DEBUG", "").split(","):  # e.g. TC_DEBUG=force_general,force_streamed (this script only)
    if key:
        name, _, value = key.partition("=")
        check(lib.tc_debug_set(name.encode(), int(value or 1)), "tc_debug_set")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(synth.CONFIGS[cfgname])
for kv in os.environ.get("TC_CFG", "").split(","):  # e.g. TC_CFG=degree=12,P=2928: overrides of the named workload
    if kv:
        k, _, v = kv.partition("=")
        cfg[k] = float(v) if "." in v else int(v)
synth.CONFIGS[cfgname] = cfg
S, T, P = cfg["S"], cfg["T"], cfg["P"]
if len(sys.argv) > 2:
    S = int(sys.argv[2])
fst = synth.config_den_fst(cfgname)
if os.environ.get("TC_PDF_PERM"):  # the same graph with its pdf ids renumbered at random (this script only)
    import numpy as np
    perm = np.random.default_rng(5).permutation(fst.num_pdfs)
    fst = synth.DenFst(fst.num_states, fst.src, fst.dst, (perm[fst.ilabel - 1] + 1).astype(fst.ilabel.dtype), fst.weight, fst.final, fst.start, fst.num_pdfs)
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P).prepare(dev)
print("graph stats", graph.stats())
y = torch.randn(S * T, P, device=dev)
deriv = torch.empty_like(y)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()


def run(with_deriv):
    rc = lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, cfg.get("l2", 0.0), 0,
        C.c_void_p(deriv.data_ptr()) if with_deriv else None, deriv.stride(0), None, None,
        C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream.cuda_stream))
    check(rc, "den")


for with_deriv in (False, True):
    for _ in range(40):  # past the clock ramp
        run(with_deriv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        run(with_deriv)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%s S=%d: %.3f ms/step  (%.2f us per frame)" % ("fwd+bwd" if with_deriv else "fwd only", S, ms, ms * 1e3 / T))
