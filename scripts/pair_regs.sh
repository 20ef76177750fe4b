#!/bin/bash
# Development aid: register / spill report of den_tied_pair.hip's <PV=1, ACCUM=false> instantiation under extra -D flags.
cd "$(dirname "$0")/../torchain_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -w -I../../include --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c den_tied_pair.hip -o /tmp/pair_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "ILi1ELb0E" | grep -E "VGPRs:|ScratchSize|Spill|SGPRs:" | tr '\n' ' '; echo
