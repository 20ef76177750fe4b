#!/bin/bash
# Development aid (runs on the GPU box): one scratch_abl/lib_<NAME>.so timed with and without a debug switch
#   scripts/abl_flag.sh NAME flag [config]
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
cp scratch_abl/lib_$1.so torchain_amd/libtorchain_hip.so
for rep in 1 2; do
  echo -n "$1 default: "; python scripts/time_den.py ${3:-C3} 2>&1 | tail -1
  echo -n "$1 $2: "; TC_DEBUG=$2 python scripts/time_den.py ${3:-C3} 2>&1 | tail -1
done
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
