#!/bin/bash
# Development aid (CPU): tc_supervision_create and tc_supervision_append (csrc/supervision.cpp, supervision_merge.cpp) under
# AddressSanitizer and UBSan over 4000 corrupted
# copies of a valid merged supervision (offsets, labels, next states, weights, final weights, S and T off by a few).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/asan_sup; mkdir -p $tmp; cd $tmp
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include -I$root/torchain_amd/csrc \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $root/scripts/asan_supervision.cpp $root/torchain_amd/csrc/supervision.cpp \
    $root/torchain_amd/csrc/supervision_merge.cpp -L/opt/rocm/lib -lamdhip64 -Wl,--unresolved-symbols=ignore-all -o fuzz_sup
python3 - <<PY
import struct, sys
sys.path.insert(0, "$root")
import numpy as np
from torchain_amd import synth
fst = synth.random_den_fst(40, 4, 24, seed=1)
sup = synth.random_supervision(fst, 3, 8, 3, seed=5, final_weights=True)
with open("$tmp/sup.bin", "wb") as f:
    f.write(struct.pack("<5if", sup.num_sequences, sup.frames_per_sequence, sup.label_dim, sup.num_states, len(sup.ilabel), sup.weight))
    for a, dt in ((sup.arc_begin, np.int32), (sup.ilabel, np.int32), (sup.nextstate, np.int32), (sup.arc_weight, np.float32), (sup.final, np.float32)):
        f.write(np.ascontiguousarray(a, dt).tobytes())
PY
LD_LIBRARY_PATH=/opt/rocm/lib ASAN_OPTIONS=detect_leaks=0 ./fuzz_sup $tmp/sup.bin 2>&1 | tail -3
