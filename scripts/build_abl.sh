#!/bin/bash
# Development aid: builds scratch_abl/lib_<NAME>.so with extra -D flags on the on-chip denominator kernels.
#   scripts/build_abl.sh NAME -DTC_PHASE_STAMPS -DTC_RESF=2 ...
set -e
cd "$(dirname "$0")/../torchain_amd/csrc"
name=$1; shift
mkdir -p ../../scratch_abl
for f in den_kernels den_tied_kernel den_tied_split den_tied_pair den_tied_mitm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -w -I../../include --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c $f.hip -o /tmp/${f}_$name.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch_abl/lib_$name.so den_graph.o den_layout.o schedule_general.o schedule_owner.o supervision.o supervision_merge.o egs_reader.o self_test.o api.o /tmp/den_kernels_$name.o /tmp/den_tied_kernel_$name.o /tmp/den_tied_split_$name.o /tmp/den_tied_pair_$name.o /tmp/den_tied_mitm_$name.o den_big_kernel.o num_kernels.o layout_kernels.o
