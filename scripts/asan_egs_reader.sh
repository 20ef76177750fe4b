#!/bin/bash
# Development aid (CPU): the native egs reader (csrc/egs_reader.cpp + supervision_merge.cpp) under AddressSanitizer and
# UBSan over truncated and byte-flipped variants of examples of every matrix / deriv-weight encoding (~22 000 inputs).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/asan_egs; mkdir -p $tmp; cd $tmp
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/include -I$root/torchain_amd/csrc \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $root/scripts/asan_egs_reader.cpp $root/torchain_amd/csrc/egs_reader.cpp \
    $root/torchain_amd/csrc/supervision_merge.cpp -o fuzz_reader
python3 - <<PY
import sys
sys.path.insert(0, "$root"); sys.path.insert(0, "$root/tests")
import kaldi_egs_writer as kw
from torchain_amd import synth
import test_egs as te
fst = synth.random_den_fst(40, 4, 24, seed=1)
for i, (kind, dw, e2e) in enumerate([("FM", "DW2", False), ("CM", "DW", True), ("CM2", None, False), ("DM", "DW2", False), ("CM3", "DW", False)]):
    eg = te.make_example(fst, 6, seed=40 + i, n_seq=1 + i % 2)
    open("$tmp/eg%d.bin" % i, "wb").write(b"\0B" + kw.chain_example(eg, matrix_kind=kind, dw=dw, e2e_flag=e2e))
PY
for i in 0 1 2 3 4; do ./fuzz_reader $tmp/eg$i.bin 2>&1 | grep -v "asan_egs_reader.cpp" | tail -2; done
