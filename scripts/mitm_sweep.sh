#!/bin/bash
# Development aid (GPU box): two CUs per sequence -- two pure recursions + combining pass (no_mitm) vs meeting in the
# middle (force_mitm), over batch sizes and graphs.
for c in "C2 1" "C2 16" "C2 32" "C2 48" "C2 64" "C2 96" "C2 128" "C5 32" "C5 64" "C5 128" "R1 32" "R1 64" "R1 128" "R2 32" "R2 64" "R3 32" "R3 64" "R3 128" "X1 32" "X1 64" "X1 128"; do
  a=$(TC_DEBUG=no_mitm python scripts/time_den.py $c 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=force_mitm python scripts/time_den.py $c 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "$c: two passes $a | meet in the middle $b"
done
