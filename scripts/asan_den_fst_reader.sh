#!/bin/bash
# Development aid (CPU): tc_den_graph_read (csrc/den_graph.cpp and the host-side schedule builder) under AddressSanitizer
# and UBSan over truncated and byte-flipped variants of a den.fst (~1600 inputs; each accepted one builds its schedules).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/asan_fst; mkdir -p $tmp; cd $tmp
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include -I$root/torchain_amd/csrc \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $root/scripts/asan_den_fst_reader.cpp $root/torchain_amd/csrc/den_graph.cpp \
    $root/torchain_amd/csrc/den_layout.cpp $root/torchain_amd/csrc/schedule_general.cpp $root/torchain_amd/csrc/schedule_owner.cpp \
    -L/opt/rocm/lib -lamdhip64 -Wl,--unresolved-symbols=ignore-all -o fuzz_fst
python3 - <<PY
import sys
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import test_abi as ta
from torchain_amd import synth
ta.write_openfst_vector("$tmp/den.fst", synth.random_den_fst(60, 4, 24, seed=1))
PY
LD_LIBRARY_PATH=/opt/rocm/lib ASAN_OPTIONS=detect_leaks=0 ./fuzz_fst $tmp/den.fst 24 2>&1 | tail -3
