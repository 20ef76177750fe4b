#!/bin/bash
# Development aid (runs on the GPU box): scripts/debug_rr.py under each scratch_abl/lib_<NAME>.so given
#   scripts/abl_debug.sh "C2 3 40" NAME...
args=$1; shift
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
for n in cur "$@"; do
  if [ $n = cur ]; then cp /tmp/cur.so torchain_amd/libtorchain_hip.so; else cp scratch_abl/lib_$n.so torchain_amd/libtorchain_hip.so; fi
  for rep in 1 2 3; do echo -n "$n: "; python scripts/debug_rr.py $args 2>&1 | grep "deriv max diff"; done
done
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
