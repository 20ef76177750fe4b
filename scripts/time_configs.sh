#!/bin/bash
# Development aid (runs on the GPU box): one line per workload and kernel family -> profiles/rNN_configs.txt
for c in C2 C3 C5 R1 R2 R3 X1 R4 X2; do echo "== $c"; python scripts/time_den.py $c 2>&1 | grep -v amdgpu.ids; done
echo "== small batches: forward and backward recursion on two CUs (default) vs the fused kernel (no_phase_split)"
for c in "C2 1" "C2 16" C2 "C2 128" C5 "R1 64" "R2 64" "R3 64" "X1 64" "X1 128"; do
  echo -n "$c two-CU: "; python scripts/time_den.py $c 2>&1 | tail -1
  echo -n "$c fused:  "; TC_DEBUG=no_phase_split python scripts/time_den.py $c 2>&1 | tail -1
done
echo "== R4 (24000 states): plane-wise on-chip kernel (default) vs the streamed path (no_planes), batch 256 / 128 / 64"
for S in 256 128 64; do
  a=$(python scripts/time_den.py R4 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_planes python scripts/time_den.py R4 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "R4 batch $S: plane-wise $a | streamed $b"
done
echo "== plane-wise kernel on random graphs of 5 / 6 / 7 planes (states x arcs per state, 2928 pdfs), batch 256"
for hd in "18000 12" "24000 12" "28000 10"; do set -- $hd
  a=$(TC_CFG=H=$1,degree=$2,P=2928 python scripts/time_den.py X2 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_planes TC_CFG=H=$1,degree=$2,P=2928 python scripts/time_den.py X2 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "H=$1 degree=$2: plane-wise $a | streamed $b"
done
echo "== split gather source (28673..40960 positions, round 6): on chip (two workgroups per sequence up to 128 sequences) vs the streamed path (no_split_source)"
for S in 256 128 64 16; do
  a=$(python scripts/time_den.py X2 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_split_source python scripts/time_den.py X2 $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "X2 batch $S: on chip $a | streamed $b"
done
for hd in "32000 10" "36000 8"; do set -- $hd
  a=$(TC_CFG=H=$1,degree=$2,P=2928 python scripts/time_den.py X2 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_split_source TC_CFG=H=$1,degree=$2,P=2928 python scripts/time_den.py X2 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "H=$1 degree=$2: on chip $a | streamed $b"
done
echo "== C3 forced general"; TC_DEBUG=force_general python scripts/time_den.py C3 2>&1 | tail -1
echo "== C3 forced streamed"; TC_DEBUG=force_streamed python scripts/time_den.py C3 2>&1 | tail -1
echo "== C3 forced streamed general"; TC_DEBUG=force_streamed,force_general python scripts/time_den.py C3 2>&1 | tail -1
echo "== C4 (2048 sequences on one GPU)"; python scripts/time_den.py C4 2>&1 | tail -2
echo "== kernel choice per graph (tc_den_graph_tuning: ms for 48 frames, one sequence per CU)"
python scripts/tuning_report.py C3 C5 R1 2>&1 | grep -v amdgpu.ids
echo "== batch curve (fwd+bwd ms): default path | fused kernel only (no_pair, no_phase_split)"
for c in C3 R1; do for S in 64 96 128 129 160 192 224 256; do
  a=$(python scripts/time_den.py $c $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_pair,no_phase_split python scripts/time_den.py $c $S 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "$c batch $S: $a | $b"
done; done
echo "== two-sequence kernel forced vs fused (batch 256)"
for c in C3 C5 R1; do
  a=$(TC_DEBUG=force_pair python scripts/time_den.py $c 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  b=$(TC_DEBUG=no_pair python scripts/time_den.py $c 2>&1 | tail -1 | grep -o "[0-9.]* ms")
  echo "$c: two-sequence $a | fused $b"
done
