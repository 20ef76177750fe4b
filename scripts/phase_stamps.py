"""Reads the in-kernel phase stamps of a -DTC_PHASE_STAMPS build (development aid)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[cfgname]
S, T, P = cfg["S"], cfg["T"], cfg["P"]
fst = synth.config_den_fst(cfgname)
H = fst.num_states
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P).prepare(dev)
print(graph.stats())
y = torch.randn(S * T, P, device=dev)
deriv = torch.empty_like(y)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()
a256 = lambda x: (x + 255) & ~255
# the stamp area is the 4 KB "scalar" block of the workspace (api.cpp: carve): the last block for batches
# beyond 128 sequences of on-chip graphs
assert S > 128, "batches of at most 128 sequences carry the second history behind the stamp area"
# api.cpp carve(): history ((T + 2) rows when the two-sequence kernel fits the graph), four double[S], two float[S], 256 B
Hs = int(os.environ.get("TC_HS", 8192))
rows = T + 2 if os.environ.get("TC_PAIR_ROOM", "1") == "1" else T + 1
off = a256(rows * S * Hs * 4) + 4 * a256(S * 8) + 2 * a256(S * 4) + 256
names = ["tail", "barrier1", "walk", "barrier2", "pass1+reduce"]
for _ in range(2):
    rc = lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, 0.0, 0,
        C.c_void_p(deriv.data_ptr()), deriv.stride(0), None, None,
        C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream.cuda_stream))
    check(rc, "den")
torch.cuda.synchronize()
st = ws[off:off + 2048].cpu().numpy().view(np.int64).reshape(2, 16, 8) / T
for ph, name in enumerate(("forward", "backward")):
    print(name, "cycles per frame; columns = waves 0..15")
    for i in range(5):
        print("  %-13s" % names[i], " ".join("%6.0f" % v for v in st[ph, :, i]))
    print("  total        ", " ".join("%6.0f" % v for v in st[ph, :, :5].sum(axis=1)))
    print("  walk: resident", " ".join("%6.0f" % v for v in st[ph, :, 5]))
    print("  walk: streamed", " ".join("%6.0f" % v for v in st[ph, :, 6]))
