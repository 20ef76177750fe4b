#!/bin/bash
# Development aid (GPU box): shader clock and power while the fused kernel runs at two batch sizes.
for S in 192 256; do
  (TC_DEBUG=no_pair,no_phase_split python - <<PY
import sys, os, ctypes as C, time
sys.path.insert(0, os.getcwd())
import torch
from torchain_amd import io, synth
from torchain_amd._lib import check, lib
for key in os.environ.get("TC_DEBUG", "").split(","):
    if key: lib.tc_debug_set(key.encode(), 1)
cfg = synth.CONFIGS["C3"]; S = $S; T, P = cfg["T"], cfg["P"]
fst = synth.config_den_fst("C3")
g = io.DenominatorGraph(fst, P).prepare(torch.device("cuda", 0))
y = torch.randn(S * T, P, device="cuda:0"); d = torch.empty_like(y)
n = lib.tc_chain_workspace_bytes(g.ptr, S, T); ws = torch.empty(n, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream()
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(200):
        lib.tc_den_forward_backward(g.ptr, S, C.c_void_p(y.data_ptr()), S * T, P, y.stride(0), 0.1, -1.0, 5e-5, 0, C.c_void_p(d.data_ptr()), d.stride(0), None, None, C.c_void_p(ws.data_ptr()), n, 0, C.c_void_p(st.cuda_stream))
    torch.cuda.synchronize()
PY
  ) > /dev/null 2>&1 &
  pid=$!
  sleep 4
  echo "== batch $S"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -6
  wait $pid
done
