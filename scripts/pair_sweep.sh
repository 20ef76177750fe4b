#!/bin/bash
# Development aid (GPU box): two-sequence kernel vs fused kernel across arcs-per-state, pdf count and batch.
for cfg in "degree=8" "degree=10" "degree=12" "degree=14" "degree=12,P=2928" "degree=8,P=2928" "degree=10,H=7168"; do
  for S in 256 192; do
    a=$(TC_CFG=$cfg python scripts/time_den.py C3 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
    b=$(TC_CFG=$cfg TC_DEBUG=force_pair python scripts/time_den.py C3 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
    echo "C3 $cfg S=$S: fused $a  pair $b"
  done
done
for c in R1; do for S in 256 224 192 160 130; do
    a=$(python scripts/time_den.py $c $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
    b=$(TC_DEBUG=force_pair python scripts/time_den.py $c $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
    echo "$c S=$S: fused $a  pair $b"
done; done
