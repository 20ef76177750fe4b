#!/bin/bash
# Development aid: builds scratch_abl/lib_<NAME>.so with extra -D flags on the denominator kernels
# (ONLY="den_tied_kernel ..." restricts the flags to those files; the others are linked as built by make).
#   scripts/build_abl.sh NAME -DTC_PHASE_STAMPS -DTC_RESF=2 ...
set -e
cd "$(dirname "$0")/../torchain_amd/csrc"
name=$1; shift
mkdir -p ../../scratch_abl
all="den_kernels den_general_owner den_tied_kernel den_tied_planes den_tied_split den_tied_pair den_tied_mitm den_slab_kernel"
only=${ONLY:-$all}
objs=""
for f in $all; do
  if [[ " $only " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -w -I../../include --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c $f.hip -o /tmp/${f}_$name.o &
    objs="$objs /tmp/${f}_$name.o"
  else
    objs="$objs $f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch_abl/lib_$name.so den_graph.o den_layout.o schedule_general.o schedule_owner.o supervision.o supervision_merge.o egs_reader.o rand_reader.o self_test.o tuning_cache.o api.o $objs num_kernels.o layout_kernels.o
