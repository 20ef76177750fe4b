#!/bin/bash
# Development aid (runs on the GPU box): times each scratch_abl/lib_<NAME>.so given on the command line on one box.
#   scripts/abl_run.sh [config] NAME...
cfg=$1; shift
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
for rep in 1 2; do
  for n in cur "$@"; do
    if [ $n = cur ]; then cp /tmp/cur.so torchain_amd/libtorchain_hip.so; else cp scratch_abl/lib_$n.so torchain_amd/libtorchain_hip.so; fi
    echo -n "$n: "; python scripts/time_den.py $cfg 2>&1 | tail -2 | tr '\n' ' '; echo
  done
done
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
