#!/bin/bash
# Development aid (GPU box): phase stamps of ab/lib_<NAME>.so (a -DTC_PHASE_STAMPS build), plane-wise kernel
cp torchain_amd/libtorchain_hip.so /tmp/cur.so
cp ab/lib_$1.so torchain_amd/libtorchain_hip.so
python scripts/phase_stamps_pw.py ${2:-R4}
cp /tmp/cur.so torchain_amd/libtorchain_hip.so
