"""Reads the raw cycle stamps of a -DTC_PAIR_STAMPS build of den_tied_pair.hip (development aid, GPU box)."""
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
dev = torch.device("cuda", 0)
check(lib.tc_debug_set(b"force_pair", 1), "force_pair")  # (the two-sequence form is opt-in)
graph = io.DenominatorGraph(fst, P).prepare(dev)
y = torch.randn(S * T, P, device=dev)
deriv = torch.empty_like(y)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream()
sb = (2 * 16 * (T + 2) * 64 + 255) & ~255   # chain_internal.h: pair_stamp_bytes, the workspace's last block
for _ in range(3):
    rc = lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, cfg.get("l2", 0.0), 0,
        C.c_void_p(deriv.data_ptr()), deriv.stride(0), None, None,
        C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream.cuda_stream))
    check(rc, "den")
torch.cuda.synchronize()
st = ws[nbytes - sb:nbytes - sb + 2 * 16 * (T + 2) * 64].cpu().numpy().view(np.int64).reshape(2, 16, T + 2, 8).astype(np.float64)
M = T // 2
fw, bw = st[0], st[1]
names_f = ["barrier1", "walk", "pass", "reduce", "tail"]
names_b = ["barrier1", "walk", "pass", "reduce", "exp(y)", "barrier3", "Y"]


def table(title, a, frames, names, nxt):
    print(title, "cycles per frame over %d frames; columns = waves 0..15" % len(frames))
    tot = np.zeros(16)
    for i, n in enumerate(names):
        d = (a[:, frames, i + 1] - a[:, frames, i]).mean(axis=1)
        tot += d
        print("  %-10s" % n, " ".join("%6.0f" % v for v in d))
    print("  %-10s" % "sum", " ".join("%6.0f" % v for v in tot))
    per = (a[:, frames[-1], 0] - a[:, frames[0], 0]) / (len(frames) - 1) * nxt
    print("  %-10s" % "frame", " ".join("%6.0f" % v for v in per))


table("forward role, first phase:", fw, list(range(2, M + 1)), names_f, 1)
table("forward role, gamma frames:", fw, list(range(M + 1, T + 1)), names_f, 1)
table("backward role, first phase:", bw, list(range(T - 2, M - 1, -1)), names_b, 1)
table("backward role, gamma frames:", bw, list(range(M - 1, 0, -1)), names_b, 1)
print("hand-off (cycles): forward role waits %.0f, backward role waits %.0f" % (
    (fw[0, T + 1, 1] - fw[0, T + 1, 0]), (bw[0, T + 1, 1] - bw[0, T + 1, 0])))
print("whole roles: forward %.0f k cycles, backward %.0f k cycles" % (
    (fw[0, T, 5] - fw[0, 1, 0]) / 1e3, (bw[0, 1, 7] - bw[0, T - 1, 0]) / 1e3))
