#!/bin/bash
# Development aid (GPU box): times one library with and without a debug switch, alternated on one box.
#   scripts/ab_flag.sh old_arrange C3 [S]
flag=$1; cfg=${2:-C3}; shift; shift
for rep in 1 2 3; do
  echo -n "with $flag: "; TC_DEBUG=no_tune,$flag python scripts/time_den.py $cfg "$@" 2>&1 | tail -1
  echo -n "default:    "; TC_DEBUG=no_tune python scripts/time_den.py $cfg "$@" 2>&1 | tail -1
done
