#!/bin/bash
# Development aid (runs on the GPU box): times several ab/lib_<NAME>.so (copied there from scratch_abl/: ab/ travels to the GPU box, scratch_abl/ does not) against the built library, alternating.
#   scripts/ab_many.sh CONFIG NAME...
cfg=$1; shift
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
for rep in $(seq ${REPS:-3}); do
  for n in CUR "$@"; do
    if [ $n = CUR ]; then cp /tmp/cur.so torchain_amd/libtorchain_hip.so; else cp ab/lib_$n.so torchain_amd/libtorchain_hip.so; fi
    echo -n "$n: "; TORCHAIN_HIP_DEBUG=no_tune python scripts/time_den.py $cfg $S 2>&1 | tail -1
  done
done
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
