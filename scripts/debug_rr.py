"""Development aid: register-row walk vs LDS-row walk of the fused tied kernel, frame by frame (alpha' history)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torchain_amd import io, synth  # noqa: E402
from torchain_amd._lib import check, lib  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = synth.CONFIGS[cfgname]
P = cfg["P"]
fst = synth.config_den_fst(cfgname)
dev = torch.device("cuda", 0)
check(lib.tc_debug_set(b"no_phase_split", 1), "dbg")
check(lib.tc_debug_set(b"no_tune", 1), "dbg")
graph = io.DenominatorGraph(fst, P).prepare(dev)
print(graph.stats())
torch.manual_seed(1)
y = torch.randn(S * T, P, device=dev)
nbytes = lib.tc_chain_workspace_bytes(graph.ptr, S, T)
stream = torch.cuda.current_stream()
out = {}
for mode in (1, 0):
    check(lib.tc_debug_set(b"reg_rows", 1 - mode), "dbg")
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    deriv = torch.zeros_like(y)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.tc_den_forward_backward(
        graph.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), cfg["leaky"], -1.0, 0.0, 0,
        C.c_void_p(deriv.data_ptr()), deriv.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(st.data_ptr()),
        C.c_void_p(ws.data_ptr()), nbytes, 0, C.c_void_p(stream.cuda_stream))
    check(rc, "den")
    torch.cuda.synchronize()
    Hs = graph.stats()["lds_bytes"] and 8192
    hist = ws[: (T + 1) * S * 8192 * 4].view(torch.float32).reshape(T + 1, S, 8192).cpu().numpy()
    out[mode] = (lp.item(), st.item(), deriv.cpu().numpy(), hist)
    print("no_reg_rows=%d logprob %.6f status %d" % (mode, lp.item(), st.item()))
a, b = out[1], out[0]
for t in range(T + 1):
    d = np.abs(a[3][t] - b[3][t])
    i = np.unravel_index(np.argmax(d), d.shape)
    print("frame %d alpha' max diff %.3e at %s (lds %.6e reg %.6e)  nbad %d" % (t, d.max(), i, a[3][t][i], b[3][t][i], (d > 1e-6 * np.abs(a[3][t]).max()).sum()))
d = np.abs(a[2] - b[2])
print("deriv max diff %.3e" % d.max())
bad = np.argwhere(np.abs(a[3][1] - b[3][1]) > 1e-6 * np.abs(a[3][1]).max())
print("bad positions frame 1 (seq, pos):", bad[:40].tolist())
if len(bad):
    pos = bad[:, 1]
    tid = (pos // 4) % 1024
    k = 4 * (pos // 4096) + pos % 4
    print("waves", np.unique(tid // 64).tolist(), "k", np.unique(k).tolist(), "lanes", np.unique(tid % 64)[:64].tolist())
dd = d.reshape(T, S, P)
per_frame = dd.max(axis=2)
for t in range(T - 1, -1, -1):
    if per_frame[t].max() > 0:
        cols = np.argwhere(dd[t] > 1e-7)
        print("frame %d: max diff per seq %s; %d bad entries; first %s" % (t, np.array2string(per_frame[t][:4], precision=2), len(cols), cols[:6].tolist()))
