"""Development aid (GPU box): a few tc_den_forward_backward calls of one workload, for counter passes
(rocprofv3 --pmc ... -- python3 scripts/den_few.py X2 [S] [calls])."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402

for key in os.environ.get("TC_DEBUG", "").split(","):
    if key:
        name, _, value = key.partition("=")
        check(lib.tc_debug_set(name.encode(), int(value or 1)), "tc_debug_set")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[cfgname]
S = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["S"]
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T, P = cfg["T"], cfg["P"]
fst = synth.config_den_fst(cfgname)
graph = io.DenominatorGraph(fst, P).prepare(torch.device("cuda", 0))
y = torch.randn(S * T, P, device="cuda:0")
deriv = torch.empty_like(y)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream()
for _ in range(calls):
    check(lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, cfg.get("l2", 0.0), 0,
        C.c_void_p(deriv.data_ptr()), deriv.stride(0), None, None, C.c_void_p(ws.data_ptr()), nbytes, 0,
        C.c_void_p(stream.cuda_stream)), "den")
torch.cuda.synchronize()
