#!/bin/bash
# Development aid: builds scratch_abl/lib_<NAME>.so with extra -D flags on den_kernels.hip.
#   scripts/build_abl.sh NAME -DTC_ABL_X ...
set -e
cd "$(dirname "$0")/../torchain_amd/csrc"
name=$1; shift
mkdir -p ../../scratch_abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -w -I../../include --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c den_kernels.hip -o /tmp/den_kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch_abl/lib_$name.so den_graph.o den_layout.o schedule_general.o schedule_owner.o supervision.o api.o /tmp/den_kernels_$name.o den_big_kernel.o num_kernels.o layout_kernels.o
