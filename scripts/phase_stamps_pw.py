"""Reads the in-kernel phase stamps of a -DTC_PHASE_STAMPS build of the plane-wise kernel (development aid).
   scripts/phase_stamps_pw.py R4"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "R4"
cfg = synth.CONFIGS[cfgname]
S, T, P = cfg["S"], cfg["T"], cfg["P"]
fst = synth.config_den_fst(cfgname)
dev = torch.device("cuda", 0)
graph = io.DenominatorGraph(fst, P).prepare(dev)
st_ = graph.stats()
print(st_)
y = torch.randn(S * T, P, device=dev)
deriv = torch.empty_like(y)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()
a256 = lambda x: (x + 255) & ~255
Hs = 4096 * ((fst.num_states + 4095) // 4096)
# the stamp area is the workspace's `scalar` block (csrc/api.cpp: carve): behind the history, the frame sums, the split-source
# scratch rows of graphs beyond 28672 positions, the per-sequence doubles and floats and the failure flag
planes = Hs // 4096
split = 0
if Hs > 28672:
    split = a256(2 * S * 4096 * (planes - (planes + 1) // 2) * 4) + a256(2 * S * (Hs + 4096 * ((planes + 1) // 2)) * 4)
off = a256((T + 1) * S * Hs * 4) + a256(S * ((T + 2 + 3) & ~3) * 4) + split + 4 * a256(S * 8) + 2 * a256(S * 4) + 256
names = ["barrier", "secondary", "plane walks", "plane passes", "reduce+tail"]
for _ in range(2):
    check(lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, 0.0, 0,
        C.c_void_p(deriv.data_ptr()), deriv.stride(0), None, None,
        C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream.cuda_stream)), "den")
torch.cuda.synchronize()
st = ws[off:off + 2048].cpu().numpy().view(np.int64).reshape(2, 16, 8) / T
for ph, name in enumerate(("forward", "backward")):
    print(name, "cycles per frame; columns = waves 0..15")
    for i in range(5):
        print("  %-13s" % names[i], " ".join("%6.0f" % v for v in st[ph, :, i]))
    print("  total        ", " ".join("%6.0f" % v for v in st[ph, :, :5].sum(axis=1)))
