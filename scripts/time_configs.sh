#!/bin/bash
# Development aid (runs on the GPU box): denominator fwd+bwd timings of every workload / kernel family.
for c in C2 C3 C5 X1 X2; do echo "== $c"; python scripts/time_den.py $c 2>&1 | grep -v amdgpu.ids; done
echo "== C3 forced general"; TC_DEBUG=force_general python scripts/time_den.py C3 2>&1 | tail -1
echo "== C3 forced streamed"; TC_DEBUG=force_streamed python scripts/time_den.py C3 2>&1 | tail -1
echo "== C3 forced streamed general"; TC_DEBUG=force_streamed,force_general python scripts/time_den.py C3 2>&1 | tail -1
echo "== C4 (2048 sequences on one GPU)"; python scripts/time_den.py C4 2>&1 | tail -2
