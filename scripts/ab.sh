#!/bin/bash
# Development aid (runs on the GPU box): times ab/lib_PREV.so against the freshly built library,
# alternating twice, same box, same clocks.  Usage: scripts/ab.sh [config] [S]   (ab/ is git-ignored but travels)
cfg=${1:-C3}; shift
cp torchain_amd/libtorchain_hip.so /tmp/new.so
for rep in 1 2; do
  cp ab/lib_PREV.so torchain_amd/libtorchain_hip.so; echo -n "prev: "; python scripts/time_den.py $cfg "$@" 2>&1 | tail -1
  cp /tmp/new.so torchain_amd/libtorchain_hip.so; echo -n "new:  "; python scripts/time_den.py $cfg "$@" 2>&1 | tail -1
done
