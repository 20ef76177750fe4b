#!/bin/bash
# Development aid (runs on the GPU box): phase stamps of scratch_abl/lib_<NAME>.so (a -DTC_PHASE_STAMPS build)
#   scripts/abl_stamps.sh NAME [config]
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
cp scratch_abl/lib_$1.so torchain_amd/libtorchain_hip.so
TORCHAIN_HIP_DEBUG=no_tune python scripts/phase_stamps.py ${2:-C3}
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
