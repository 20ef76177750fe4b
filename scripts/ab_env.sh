#!/bin/bash
# Development aid (runs on the GPU box): times one library under several values of an environment variable, alternating.
#   scripts/ab_env.sh CONFIG VAR VALUE...      (the value "-" leaves the variable unset)
cfg=$1; var=$2; shift 2
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    echo -n "$var=$v: "
    if [ "$v" = "-" ]; then TORCHAIN_HIP_DEBUG=no_tune python scripts/time_den.py $cfg 2>&1 | tail -1
    else env $var=$v TORCHAIN_HIP_DEBUG=no_tune python scripts/time_den.py $cfg 2>&1 | tail -1; fi
  done
done
